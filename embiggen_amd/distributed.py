"""One-process-per-GPU data parallelism for the SkipGram training path (torch.distributed; backend
"nccl" = RCCL over xGMI on the GPU box, "gloo" in the CPU tests).

The reference has no counterpart: ensmallen parallelises with rayon threads inside one process
(SURVEY.md section 2a).  Units (walks) are partitioned by walk id; the exchange step is the
block-partitioned scheme below, chosen after measuring that schemes which let several GPUs move
the same row between exchanges (replica delta-sum, row caches with delta scatter) lose or destroy
embedding quality on scale-free graphs.
"""
from typing import Tuple


def walk_slice(step: int, rank: int, world: int, walks_per_step: int) -> Tuple[int, int]:
    """(first_walk_id, n_walks) trained by `rank` in global step `step`: slices are disjoint
    across ranks and steps and cover the walk ids contiguously."""
    return (step * world + rank) * walks_per_step, walks_per_step


class TorchComm:
    """torch.distributed as the exchange fabric (backend "nccl" = RCCL on the GPU box)."""

    def __init__(self, group=None):
        import torch.distributed as dist

        self._dist, self._group = dist, group
        self.rank = dist.get_rank(group)
        self.world = dist.get_world_size(group)

    def exchange_counts(self, counts):
        """counts[r] = rows this rank sends to rank r  ->  rows it receives from every rank."""
        import torch

        out = torch.empty_like(counts)
        self._dist.all_to_all_single(out, counts, group=self._group)
        return out

    def exchange_rows(self, rows, send_counts, recv_counts):
        """All-to-all of row blocks: rows is grouped by destination rank (send_counts rows each);
        returns the received blocks concatenated in source-rank order."""
        import torch

        out = torch.empty((int(sum(recv_counts)),) + tuple(rows.shape[1:]), dtype=rows.dtype,
                          device=rows.device)
        self._dist.all_to_all_single(out, rows.contiguous(), output_split_sizes=list(recv_counts),
                                     input_split_sizes=list(send_counts), group=self._group)
        return out


class LoopbackComm:
    """world = 1: everything is local."""

    rank, world = 0, 1

    def exchange_counts(self, counts):
        return counts.clone()

    def exchange_rows(self, rows, send_counts, recv_counts):
        return rows


# ---------------------------------------------------------------------------------------------
# Block-partitioned trainer: conflict-free multi-GPU SkipGram.
#
# Measured on one GPU with simulated ranks (scripts/replica_quality.py, sharded_quality2.py):
# merging the displacements of replicas / row caches that several ranks moved at once is unstable
# on scale-free graphs (8 replicas: link AUROC 0.02-0.27 where one trainer reaches 0.98), because
# hub rows move far within one exchange interval.  The robust form shares no row between GPUs at
# any time (GraphVite's orthogonal blocks): nodes are partitioned p(v) = v % world; GPU i owns
# central partition i for good and holds ONE context partition at a time, which rotates around a
# ring.  A (centre, context) pair is trained on the GPU that owns its centre, in the episode in
# which that GPU holds the pair's context partition; negatives come from the resident context
# partition (degree-proportional within it).  After `world` episodes every block (i, j) of the
# round's pairs has been trained exactly once and every partition is home again.  Traffic per
# round and GPU: its share of the pair list (8 B/pair) + world rotations of N/world context rows.
# ---------------------------------------------------------------------------------------------


def partition_rows(n_nodes: int, part: int, world: int) -> int:
    return (n_nodes - part + world - 1) // world


class BlockPartitionedTrainer:
    def __init__(self, graph, train_params, d: int, ld: int, seed: int, init_scale: float, comm,
                 device, scale_free: bool = True, init_fn=None, compute=None):
        import torch

        self.graph, self.tp, self.comm = graph, train_params, comm
        self.device, self.ld, self.d = torch.device(device), ld, d
        self.n_nodes = graph.get_number_of_nodes()
        rank, world = comm.rank, comm.world
        if init_fn is None:
            from . import ops

            def init_fn(table_id):
                return ops.init_table(self.n_nodes, d, seed, table_id, init_scale,
                                      device=self.device.index or 0, ld=ld)
        self.central = init_fn(0)[rank::world].contiguous()   # partition `rank`, never moves
        self.context = init_fn(1)[rank::world].contiguous()   # resident context partition
        self.resident = rank
        if getattr(graph, "_device_tensors", None) is not None:
            col = graph._device_tensors["col_idx"].to(torch.int64) & 0xFFFFFFFF
        else:
            col = torch.from_numpy(graph.col_idx.astype("int64")).to(self.device)
        self.pools = []
        for p in range(world):
            if scale_free:
                rows = torch.div(col[col % world == p], world, rounding_mode="floor")
            else:
                rows = torch.arange(partition_rows(self.n_nodes, p, world), device=self.device)
            if rows.numel() == 0:
                raise ValueError("A partition owns no edge endpoint: graph too small to split.")
            self.pools.append(rows.to(torch.int32).contiguous())
        self.compute = compute or self._gpu_compute
        self.pairs_seen = 0  # rounds completed (keys the pair ids)
        self.last_round = None

    def _gpu_compute(self, pairs, rows, part, seed, epoch, first_pair, lr):
        from . import ops

        ops.step(self.graph, self.tp, pairs, seed, epoch, first_pair, lr, self.central,
                 self.context, walk_rows=rows, neg_pool=self.pools[part],
                 neg_id_mul=self.comm.world, neg_id_add=part, pair_mode=True)

    def _rotate(self):
        """Send the resident context partition to the previous rank, receive the next one's."""
        comm = self.comm
        world = comm.world
        nxt = (self.resident + 1) % world
        send = [0] * world
        recv = [0] * world
        send[(comm.rank - 1) % world] = self.context.shape[0]
        recv[(comm.rank + 1) % world] = partition_rows(self.n_nodes, nxt, world)
        self.context = comm.exchange_rows(self.context, send, recv)
        self.resident = nxt

    def train_round(self, walks, window: int, min_dist: int, seed: int, epoch: int, lr: float,
                    pairs=None):
        """One round = every (centre, context) pair of this rank's walks (or the explicit
        ``pairs`` int32 [n, 2]), plus the pairs the other ranks route here, trained block by
        block while the context partitions go round the ring."""
        import torch

        comm = self.comm
        world = comm.world
        if pairs is None:
            # fused path: the pair kernel also emits the sort key of every slot
            from . import ops

            slots, keys = ops.walk_pair_blocks(
                walks, window, min_dist, world,
                (seed * 0x9E3779B97F4A7C15 + self.pairs_seen * world + comm.rank) & (2 ** 63 - 1))
            keys, order = torch.sort(keys)
            bounds = torch.arange(world * world + 1, dtype=torch.int64, device=keys.device) << 31
            edges = torch.searchsorted(keys, bounds)
            n = int(edges[-1])  # unused slots carry INT64_MAX and sort behind every block
            counts = (edges[1:] - edges[:-1]).reshape(world, world)
            sorted_pairs = slots[order[:n]]
            del slots, keys, order
        else:
            # explicit pairs (tests, CPU): same key built with integer tensor ops, so CPU and GPU
            # tensors sort identically.  The salt shuffles pairs inside a block: pairs leave the
            # walks ~10 in a row with the same centre, and one wavefront per pair would otherwise
            # make them hammer that row at the same time.
            n = pairs.shape[0]
            centre = pairs[:, 0].to(torch.int64) & 0xFFFFFFFF
            ctx = pairs[:, 1].to(torch.int64) & 0xFFFFFFFF
            block = (centre % world) * world + ctx % world
            idx = torch.arange(n, dtype=torch.int64, device=pairs.device)
            salt = (idx * 0x3C6EF35F + (seed * 0x19660D + self.pairs_seen * 0x2545F491 + 1)) & 0x7FFFFFFF
            salt = ((salt ^ (salt >> 15)) * 0x2C1B3C6D) & 0x7FFFFFFF
            salt = ((salt ^ (salt >> 12)) * 0x297A2D39) & 0x7FFFFFFF
            salt = salt ^ (salt >> 15)
            order = torch.argsort(block * (1 << 31) + salt)
            sorted_pairs = pairs[order]
            # counts[o][p] = my pairs for owner o with context partition p; owner o gets row o
            counts = torch.bincount(block, minlength=world * world).reshape(world, world)
            del idx, salt, centre, ctx, order, block
        got_counts = comm.exchange_rows(counts, [1] * world, [1] * world)  # [source][partition]
        send_l = counts.sum(1).tolist()
        src_part = got_counts.tolist()
        recv_l = [sum(row) for row in src_part]
        mine = comm.exchange_rows(sorted_pairs, send_l, recv_l)
        del sorted_pairs
        # block j = the j-th sub-segment of every source's segment (no second sort needed)
        seg_start, pieces, sizes = 0, [[] for _ in range(world)], [0] * world
        for src in range(world):
            off = seg_start
            for j in range(world):
                c = src_part[src][j]
                if c:
                    pieces[j].append(mine[off:off + c])
                    sizes[j] += c
                off += c
            seg_start = off
        flat = [piece for j in range(world) for piece in pieces[j]]
        blocks = torch.cat(flat) if len(flat) > 1 else (flat[0] if flat else mine[:0])
        starts = [0]
        for s in sizes:
            starts.append(starts[-1] + s)
        rows = torch.div(blocks.to(torch.int64) & 0xFFFFFFFF, world,
                         rounding_mode="floor").to(torch.int32).contiguous()
        # distinct RNG keys for every pair ever trained on any rank
        base = (self.pairs_seen * world + comm.rank) << 32
        for _ in range(world):
            j = self.resident
            lo, hi = starts[j], starts[j + 1]
            if hi > lo:
                self.compute(blocks[lo:hi], rows[lo:hi], j, seed, epoch, base + lo, lr)
            if world > 1:
                self._rotate()
        self.pairs_seen += 1
        self.last_round = {"pairs_generated": int(n), "pairs_trained": int(mine.shape[0]),
                           "block_sizes": sizes}

    def gather_full(self):
        """(central, contextual) as full [N, ld] tables on every rank."""
        import torch

        comm, world = self.comm, self.comm.world
        out = []
        for shard, part in ((self.central, comm.rank), (self.context, self.resident)):
            if world == 1:
                out.append(shard.clone())
                continue
            n_max = partition_rows(self.n_nodes, 0, world)
            padded = torch.zeros((n_max, self.ld), dtype=shard.dtype, device=shard.device)
            padded[: shard.shape[0]] = shard
            blocks = comm.exchange_rows(padded.repeat(world, 1), [n_max] * world, [n_max] * world)
            parts = comm.exchange_counts(torch.full((world,), part, dtype=torch.int64,
                                                    device=shard.device)).tolist()
            full = torch.empty((self.n_nodes, self.ld), dtype=shard.dtype, device=shard.device)
            for r in range(world):
                p = parts[r]
                full[p::world] = blocks[r * n_max: r * n_max + partition_rows(self.n_nodes, p, world)]
            out.append(full)
        return out

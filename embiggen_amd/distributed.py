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
#
# Centre records: a wave that trains one (centre, context) pair reads and writes the centre row
# for that single pair.  The pairs of a block are therefore grouped by centre node and packed
# into records [centre, up to C contexts]: the centre row stays in registers over the record and
# the sample rounds are packed -- measured 0.61 -> 0.69 of the HBM roofline on one GPU (+13 %),
# and 4.4 B instead of 8 B per pair on the wire.  Records, not pairs, are shuffled inside a block.
# ---------------------------------------------------------------------------------------------


def partition_rows(n_nodes: int, part: int, world: int) -> int:
    return (n_nodes - part + world - 1) // world


class BlockPartitionedTrainer:
    def __init__(self, graph, train_params, d: int, ld: int, seed: int, init_scale: float, comm,
                 device, scale_free: bool = True, init_fn=None, compute=None,
                 record_contexts: int = 10):
        import torch

        self.graph, self.tp, self.comm = graph, train_params, comm
        self.record_contexts = int(record_contexts)
        # records of 1 + C ids are trained with window = C (only position 0 is a centre)
        self.tp_records = type(train_params).from_buffer_copy(train_params)
        self.tp_records.window, self.tp_records.min_dist = self.record_contexts, 1
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

        tp = self.tp if pairs.shape[1] == 2 else self.tp_records
        ops.step(self.graph, tp, pairs, seed, epoch, first_pair, lr, self.central,
                 self.context, walk_rows=rows, neg_pool=self.pools[part],
                 neg_id_mul=self.comm.world, neg_id_add=part, pair_mode=True)

    def _pack_records(self, pairs, keys, seed: int):
        """pairs int32 [n, 2] sorted by keys = block << 32 | centre  ->  (records int32
        [R, 1 + C] grouped by block and shuffled inside a block, counts int64 [world, world]):
        every run of equal (block, centre) is cut into records of up to C contexts."""
        import torch

        C, world, dev = self.record_contexts, self.comm.world, pairs.device
        n = pairs.shape[0]
        if n == 0:
            return (torch.empty((0, 1 + C), dtype=torch.int32, device=dev),
                    torch.zeros((world, world), dtype=torch.int64, device=dev))
        start = torch.ones(n, dtype=torch.bool, device=dev)
        torch.ne(keys[1:], keys[:-1], out=start[1:])
        run_pos = torch.nonzero(start).flatten()
        run_id = torch.cumsum(start, 0) - 1
        del start
        rank = torch.arange(n, dtype=torch.int64, device=dev) - run_pos[run_id]
        run_len = torch.diff(run_pos, append=torch.tensor([n], dtype=torch.int64, device=dev))
        recs = torch.div(run_len + (C - 1), C, rounding_mode="floor")
        first_rec = torch.cumsum(recs, 0) - recs
        rec_id = first_rec[run_id] + torch.div(rank, C, rounding_mode="floor")
        del run_id, run_len, first_rec
        n_rec = int(recs.sum())
        records = torch.full((n_rec, 1 + C), -1, dtype=torch.int32, device=dev)
        flat = records.view(-1)
        flat[rec_id * (1 + C)] = pairs[:, 0]
        flat[rec_id * (1 + C) + 1 + rank % C] = pairs[:, 1]
        del rec_id, rank
        rec_block = torch.repeat_interleave(keys[run_pos] >> 32, recs)
        del run_pos, recs
        counts = torch.bincount(rec_block, minlength=world * world).reshape(world, world)
        # hub centres own thousands of consecutive records: shuffle the records of a block so
        # that concurrent waves do not all accumulate into the same centre row
        idx = torch.arange(n_rec, dtype=torch.int64, device=dev)
        salt = (idx * 0x3C6EF35F + (seed * 0x19660D + self.pairs_seen * 0x2545F491 + 1)) & 0x7FFFFFFF
        salt = ((salt ^ (salt >> 15)) * 0x2C1B3C6D) & 0x7FFFFFFF
        salt = ((salt ^ (salt >> 12)) * 0x297A2D39) & 0x7FFFFFFF
        order = torch.argsort((rec_block << 31) | (salt ^ (salt >> 15)), stable=True)
        del idx, salt, rec_block
        return records[order], counts

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
            # fused path: the pair kernel also emits the sort key of every slot (block << 32 |
            # centre); one sort groups the pairs by block and centre, then they are packed into
            # centre records
            from . import ops

            slots, keys = ops.walk_pair_blocks(walks, window, min_dist, world, 2 ** 64 - 1)
            keys, order = torch.sort(keys, stable=True)  # ties keep the slot order
            last = torch.tensor([world * world << 32], dtype=torch.int64, device=keys.device)
            n = int(torch.searchsorted(keys, last)[0])  # unused slots carry INT64_MAX
            grouped = slots[order[:n]]
            del slots, order
            sorted_pairs, counts = self._pack_records(grouped, keys[:n], seed)
            del grouped, keys
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
        if self.n_nodes < 2 ** 31:  # ids are non-negative as int32 (sentinels stay negative)
            rows = torch.div(blocks, world, rounding_mode="trunc")
        else:
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
        self.last_round = {"pairs_generated": int(n),
                           "pairs_trained": int((mine[:, 1:] != -1).sum()),
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

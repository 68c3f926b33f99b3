"""One-process-per-GPU data parallelism for the training path (torch.distributed; backend
"nccl" = RCCL over xGMI on the GPU box, "gloo" in the CPU tests).

Round-1 form: the CSR and both tables are replicated, walks are partitioned
by walk id, and after every step the replicas exchange what they learned.  Because every rank only
touches the rows its walks visit, replicas are combined by **summing deltas**
    new = base + sum_r (table_r - base)  =  all_reduce_sum(table_r) - (world - 1) * base
rather than by averaging (averaging would divide the update of a row only one rank touched by the
world size).  One all-reduce per table per step: 2 x 4*N*ld bytes per 1250 * walks_per_step pairs.
The reference has no counterpart: ensmallen parallelises with rayon threads inside one process
(SURVEY.md section 2a).
"""
from typing import Tuple


def walk_slice(step: int, rank: int, world: int, walks_per_step: int) -> Tuple[int, int]:
    """(first_walk_id, n_walks) trained by `rank` in global step `step`: slices are disjoint
    across ranks and steps and cover the walk ids contiguously."""
    return (step * world + rank) * walks_per_step, walks_per_step


class ReplicaSync:
    """Keeps the last agreed copy of each table and folds every rank's delta into it."""

    def __init__(self, *tables):
        import torch.distributed as dist

        self._dist = dist
        self.world = dist.get_world_size() if dist.is_initialized() else 1
        self.tables = tables
        self.bases = [t.clone() for t in tables] if self.world > 1 else []

    def sync(self):
        """Blocking on the current stream: after it every rank holds base + sum of all deltas."""
        if self.world == 1:
            return
        for table, base in zip(self.tables, self.bases):
            self._dist.all_reduce(table, op=self._dist.ReduceOp.SUM)
            table.add_(base, alpha=-(self.world - 1))
            base.copy_(table)


# ---------------------------------------------------------------------------------------------
# Row-sharded tables: the north-star form.  owner(v) = v % world, local row = v // world.
# Per batch of walks a rank (1) collects the distinct nodes its walks visit, (2) fetches their rows
# of both tables from the owners into compact row caches (all-to-all), (3) runs the fused kernel on
# the caches with negatives drawn from -- and updated in place in -- its own shard, and (4) sends
# the cache deltas back to the owners, which add them (all-to-all).  Only the needed rows travel:
# <= walk_length rows per 2wL - w(w+1) pairs, ~184 B/pair against 12 288 B/pair of HBM traffic
# (SURVEY.md section 8e), so xGMI is never the limiter, and no ring over a whole table exists.
# ---------------------------------------------------------------------------------------------


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


class RowShardedTables:
    """This rank's rows of both tables plus its pool of negative rows."""

    def __init__(self, graph, d: int, ld: int, seed: int, init_scale: float, comm, device,
                 scale_free: bool = True, init_fn=None):
        import torch

        self.comm, self.device, self.ld, self.d = comm, torch.device(device), ld, d
        self.n_nodes = graph.get_number_of_nodes()
        rank, world = comm.rank, comm.world
        self.n_local = (self.n_nodes - rank + world - 1) // world
        if init_fn is None:
            from . import ops

            def init_fn(table_id):
                return ops.init_table(self.n_nodes, d, seed, table_id, init_scale,
                                      device=self.device.index or 0, ld=ld)
        self.central = init_fn(0)[rank::world].contiguous()
        self.contextual = init_fn(1)[rank::world].contiguous()
        assert self.central.shape == (self.n_local, ld)
        if scale_free:
            if getattr(graph, "_device_tensors", None) is not None:
                col = graph._device_tensors["col_idx"].to(torch.int64) & 0xFFFFFFFF
            else:
                col = torch.from_numpy(graph.col_idx.astype("int64")).to(self.device)
            mine = col[col % world == rank]
            self.neg_pool = torch.div(mine, world, rounding_mode="floor").to(torch.int32)
        else:
            self.neg_pool = torch.arange(self.n_local, dtype=torch.int32, device=self.device)
        if self.neg_pool.numel() == 0:
            raise ValueError("This rank owns no edge endpoint: the graph is too small to shard.")

    def gather_full(self):
        """(central, contextual) as full [N, ld] tables on every rank (result extraction)."""
        import torch

        out = []
        for shard in (self.central, self.contextual):
            world = self.comm.world
            if world == 1:
                out.append(shard.clone())
                continue
            n_max = (self.n_nodes + world - 1) // world
            padded = torch.zeros((n_max, self.ld), dtype=shard.dtype, device=shard.device)
            padded[: shard.shape[0]] = shard
            blocks = self.comm.exchange_rows(padded.repeat(world, 1), [n_max] * world,
                                             [n_max] * world)
            full = torch.empty((self.n_nodes, self.ld), dtype=shard.dtype, device=shard.device)
            for r in range(world):
                n_r = (self.n_nodes - r + world - 1) // world
                full[r::world] = blocks[r * n_max: r * n_max + n_r]
            out.append(full)
        return out


class ShardedTrainer:
    """One rank of the row-sharded trainer.  ``compute(walks, rows, cache_central,
    cache_contextual, tables)`` runs the fused kernel on the row caches (default: ``ops.step``)."""

    def __init__(self, graph, tables: RowShardedTables, train_params, compute=None,
                 merge: str = "mean"):
        if merge not in ("mean", "sum"):
            raise ValueError("merge must be 'mean' or 'sum'")
        self.merge = merge
        self.graph, self.tables, self.tp = graph, tables, train_params
        self.compute = compute or self._gpu_compute
        self.last_exchange = None

    def _gpu_compute(self, walks, rows, cache_c, cache_x, seed, epoch, first_walk, lr):
        from . import _lib, ops

        t = self.tables
        negative = t.central if self.tp.model == _lib.MODEL_CBOW else t.contextual
        ops.step(self.graph, self.tp, walks, seed, epoch, first_walk, lr, cache_c, cache_x,
                 walk_rows=rows, negative=negative, neg_pool=t.neg_pool,
                 neg_id_mul=t.comm.world, neg_id_add=t.comm.rank)

    def train_batch(self, walks, seed: int, epoch: int, first_walk: int, lr: float):
        """walks: int32 [n_walks, walk_length] of global node ids (uint32 bits) on the device."""
        import torch

        t, comm = self.tables, self.tables.comm
        world, ld = comm.world, t.ld
        ids = walks.to(torch.int64).flatten() & 0xFFFFFFFF
        uniq, inv = torch.unique(ids, return_inverse=True)
        if uniq[-1] == 0xFFFFFFFF:  # walk padding after a trap node: never addressed
            uniq = uniq[:-1]
        rows = inv.to(torch.int32).reshape(walks.shape).contiguous()

        owner = uniq % world
        order = torch.argsort(owner, stable=True)
        counts = torch.bincount(owner, minlength=world)
        recv_counts = comm.exchange_counts(counts)
        send_l, recv_l = counts.tolist(), recv_counts.tolist()
        wanted = torch.div(uniq[order], world, rounding_mode="floor")
        requested = comm.exchange_rows(wanted, send_l, recv_l)  # local rows others (and I) need

        payload = torch.cat([t.central[requested], t.contextual[requested]], dim=1)
        got = comm.exchange_rows(payload, recv_l, send_l)
        cache = torch.empty_like(got)
        cache[order] = got
        cache_c, cache_x = cache[:, :ld].contiguous(), cache[:, ld:].contiguous()
        old_c, old_x = cache_c.clone(), cache_x.clone()

        self.compute(walks, rows, cache_c, cache_x, seed, epoch, first_walk, lr)

        delta = torch.cat([cache_c - old_c, cache_x - old_x], dim=1)[order]
        back = comm.exchange_rows(delta, send_l, recv_l)
        if self.merge == "mean":
            # a row cached by several ranks comes back as several displacements of the same base:
            # average them (summing overshoots once a row moves appreciably within a batch, which
            # hub rows always do); rows held by one rank keep their full update
            holders = torch.bincount(requested, minlength=t.n_local).to(back.dtype)
            back = back / holders[requested].unsqueeze(1)
        t.central.index_add_(0, requested, back[:, :ld])
        t.contextual.index_add_(0, requested, back[:, ld:])
        self.last_exchange = {"unique_nodes": int(uniq.numel()), "rows_sent": int(sum(send_l)),
                              "rows_served": int(sum(recv_l))}


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
            from . import ops

            pairs = ops.walk_pairs(walks, window, min_dist)
        centre = pairs[:, 0].to(torch.int64) & 0xFFFFFFFF
        owner = centre % world
        order = torch.argsort(owner, stable=True)
        counts = torch.bincount(owner, minlength=world)
        recv_counts = comm.exchange_counts(counts)
        mine = comm.exchange_rows(pairs[order], counts.tolist(), recv_counts.tolist())

        ctx = mine[:, 1].to(torch.int64) & 0xFFFFFFFF
        part = ctx % world
        by_block = torch.argsort(part, stable=True)
        blocks = mine[by_block].contiguous()
        sizes = torch.bincount(part, minlength=world).tolist()
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
        self.last_round = {"pairs_generated": int(pairs.shape[0]), "pairs_trained": int(mine.shape[0]),
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

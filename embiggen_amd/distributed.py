"""One-process-per-GPU training of the SkipGram path (torch.distributed; backend "nccl" = RCCL over
xGMI on the GPU box, "gloo" in the CPU tests).

The reference has no counterpart: ensmallen parallelises with rayon threads inside one process
(SURVEY.md section 2a); the call being spread over the GPUs is
``self._model.fit_transform(graph)`` (embedders/ensmallen_embedders/node2vec.py:99).

Scheme (chosen after measuring that every scheme which lets two GPUs move the
same row between exchanges loses or destroys embedding quality on scale-free graphs): no row is
ever held by two GPUs.

* Central table: striped over the ranks, ``owner(c) = c % world``, row ``c // world``; a rank's
  partition never moves.
* Contextual table: striped into ``parts = P * world`` parts, P >= 2 (``x % parts``, row
  ``x // parts``).  A rank holds P parts at a time.  In global episode ``g`` rank ``r`` trains part
  ``(P r + g) % parts``; while it does, the part it finished in episode ``g - 1`` travels to rank
  ``r - 1`` (which needs it in episode ``g + P - 1``) and the part for episode ``g + P - 1``
  arrives from rank ``r + 1`` -- every transfer is hidden behind the training of other parts
  (RCCL send / receive on its own stream); P + 1 part buffers per rank.
* Pairs: a round = ``round_walks`` walks per rank.  The walks (u32 ids, 512 B each) are
  all-gathered; every rank extracts from ALL walks of the round the (centre, context) pairs whose
  centre it owns, sorted by (context part, centre) -- two passes over the walks and one radix sort
  in HIP (``gn2v_block_count`` / ``gn2v_block_extract``); nothing else crosses the fabric.
* Training: ``gn2v_block_step`` per part, negatives drawn degree-proportionally inside the
  cell of the pair's context (one alias table per cell).

Preparation of round ``t + 1`` (walk generation, all-gather, extraction, sort) runs on a second
stream while round ``t`` trains.
"""
import os
import sys
from typing import List, Optional, Tuple


def walk_slice(step: int, rank: int, world: int, walks_per_step: int) -> Tuple[int, int]:
    """(first_walk_id, n_walks) trained by `rank` in global step `step`: slices are disjoint
    across ranks and steps and cover the walk ids contiguously."""
    return (step * world + rank) * walks_per_step, walks_per_step


def stripe_rows(n: int, first: int, stride: int) -> int:
    """Number of ids first, first + stride, ... below n."""
    return (n - first + stride - 1) // stride if n > first else 0


PAIR_ROOM = 1 << 24  # pair buffers are sized in steps of this many pairs
MIN_ROWS_PER_CELL = 32768


def round_walks_within(free_bytes: int, walk_length: int, window: int, key_bits: int, world: int,
                       overlap: bool) -> int:
    """Walks per rank and round for ``free_bytes`` of HBM (``gn2v_block_round_walks``: the largest
    power of two <= 2^23 whose pair buffers -- held once by the round in training, twice by the
    round being built -- fit three quarters of it).  Every rank must use the same value."""
    import ctypes as C

    from . import _lib

    out = C.c_uint64()
    _lib.check(_lib.lib().gn2v_block_round_walks(int(free_bytes), walk_length, window, key_bits,
                                                 world, int(bool(overlap)), C.byref(out)))
    return out.value


def auto_plan(n_nodes: int, world: int) -> Tuple[int, int]:
    """(parts, slices) of the contextual table (``gn2v_block_auto_plan``: one rule for the C++
    one-GPU fit and for this trainer).  Measured (scripts/quality_probe.py):
    the smaller a cell, the more of it lives in the XCD's L2 (BA 10 M nodes: 0.67 of the HBM
    roofline unsliced, 0.80 at 4 x 8 cells, 0.87 at 32 x 8) -- and the more often two waves
    read-modify-write the same row at once; link quality stays at or above the walk-ordered
    trainer's while a cell keeps >= 32 k rows.  slices = 8 (one per XCD); as many parts -- at
    least two per rank with several ranks, at most 128 -- as keep MIN_ROWS_PER_CELL rows a cell."""
    import ctypes as C

    from . import _lib

    parts, slices = C.c_uint32(), C.c_uint32()
    _lib.check(_lib.lib().gn2v_block_auto_plan(n_nodes, world, C.byref(parts), C.byref(slices)))
    return parts.value, slices.value


class _Done:
    def wait(self):
        return None


class LoopbackComm:
    """world = 1: everything is local."""

    rank, world = 0, 1
    backend = "loopback"

    def all_gather(self, tensor):
        return tensor

    def sendrecv_start(self, send, dst, recv, src):
        recv.copy_(send)
        return _Done()


class _Works:
    def __init__(self, works, after=None):
        self._works, self._after = works, after

    def wait(self):
        for w in self._works:
            w.wait()
        if self._after is not None:
            self._after()


class TorchComm:
    """torch.distributed as the exchange fabric (backend "nccl" = RCCL on the GPU box).  With the
    "gloo" backend (tests) device tensors are staged through host memory."""

    def __init__(self, group=None):
        import torch.distributed as dist

        self._dist, self._group = dist, group
        self.rank = dist.get_rank(group)
        self.world = dist.get_world_size(group)
        self.backend = str(dist.get_backend(group))
        self._staged = self.backend == "gloo"
        # the walk all-gather of round t + 1 must not queue behind the part rotations of round t:
        # it gets a communicator (and with RCCL a stream) of its own
        self._gather_group = group
        if self.world > 1:
            ranks = dist.get_process_group_ranks(group) if group is not None else None
            self._gather_group = dist.new_group(ranks=ranks, backend=self.backend)

    def all_gather(self, tensor):
        """[n, ...] on every rank -> [world * n, ...] in rank order."""
        import torch

        t = tensor.contiguous()
        if self._staged and t.is_cuda:
            host = t.cpu()
            out = torch.empty((self.world * host.shape[0],) + tuple(host.shape[1:]), dtype=host.dtype)
            self._dist.all_gather_into_tensor(out, host, group=self._gather_group)
            return out.to(t.device)
        out = torch.empty((self.world * t.shape[0],) + tuple(t.shape[1:]), dtype=t.dtype,
                          device=t.device)
        self._dist.all_gather_into_tensor(out, t, group=self._gather_group)
        return out

    def sendrecv_start(self, send, dst, recv, src):
        """Start sending `send` to rank `dst` and receiving `recv` from rank `src`; returns a
        handle whose ``wait()`` orders the caller's stream (host, for gloo) after both."""
        dist = self._dist
        if self._staged and send.is_cuda:
            host_out, host_in = send.cpu(), recv.cpu()
            works = dist.batch_isend_irecv([dist.P2POp(dist.isend, host_out, dst, self._group),
                                            dist.P2POp(dist.irecv, host_in, src, self._group)])
            return _Works(works, after=lambda: recv.copy_(host_in))
        works = dist.batch_isend_irecv([dist.P2POp(dist.isend, send, dst, self._group),
                                        dist.P2POp(dist.irecv, recv, src, self._group)])
        return _Works(works)


class GpuBlockBackend:
    """Device side of the trainer: thin calls into the C ABI (``ops``)."""

    def __init__(self, graph, device):
        import torch

        self.graph, self.device = graph, torch.device(device)
        self.index = self.device.index or 0
        self._slots, self._temp = {}, None  # standing pair buffers (prepare(slot=...))

    def init_rows(self, n_rows, d, ld, seed, table_id, scale, first_row, stride, out=None):
        from . import ops

        return ops.init_table_rows(n_rows, d, seed, table_id, scale, first_row, stride,
                                   device=self.index, ld=ld, out=out)

    def empty_rows(self, n_rows, ld):
        import torch

        return torch.empty((n_rows, ld), dtype=torch.float32, device=self.device)

    def plan(self, **kw):
        from . import ops

        return ops.block_plan(self.graph, device=self.index, **kw)

    def alias_tables(self, plan):
        from . import ops

        return ops.block_alias(self.graph, plan, device=self.index)

    def prepare(self, plan, walks_all, seed, epoch, first_walk, hub_bits=None, scale=1.0,
                slot=None):
        """-> (keys, vals, cell_offsets, n_pairs); one host read (the pair count).

        ``slot`` None: fresh buffers of this round's size.  ``slot`` 0 / 1: the trainer's standing
        buffers -- one (keys, vals) pair per slot and one sort buffer, allocated once for
        ``scale`` times the pairs of this round (a short first or last round of a fit is sized
        like the full ones) and kept until ``release()``: 2 x 42 GB + 84 GB per round of 2^22
        walks must not depend on which cached block the allocator happens to split."""
        from . import ops

        import torch

        work, offsets = ops.block_count(self.graph, plan, walks_all, seed, epoch, first_walk)
        n_pairs = int(offsets[-1])
        dev = walks_all.device
        key_type = torch.int64 if plan.key_bits == 64 else torch.int32

        def rounded(n):  # in steps of 2^24 pairs
            return max(1, -(-int(n) // PAIR_ROOM)) * PAIR_ROOM

        if slot is None:
            room = rounded(n_pairs)
            keys = torch.empty(room, dtype=key_type, device=dev)
            vals = torch.empty(room, dtype=torch.int32, device=dev)
            temp = torch.empty(ops.block_extract_temp_bytes(room, plan.key_bits),
                               dtype=torch.uint8, device=dev)
        else:
            held = self._slots.get(slot)
            if held is None or held[0].numel() < n_pairs or held[0].dtype != key_type:
                self._slots[slot] = held = None  # released before its successor is allocated
                # the rounds of a fit differ by a fraction of a percent: 1/64 of head room
                room = rounded(n_pairs * max(1.0, scale) * (1 + 1 / 64))
                held = self._slots[slot] = (torch.empty(room, dtype=key_type, device=dev),
                                            torch.empty(room, dtype=torch.int32, device=dev))
            keys, vals = held
            need = ops.block_extract_temp_bytes(keys.numel(), plan.key_bits)
            if self._temp is None or self._temp.numel() < need:
                self._temp = None
                self._temp = torch.empty(need, dtype=torch.uint8, device=dev)
            temp = self._temp
        ops.block_extract(self.graph, plan, walks_all, seed, epoch, first_walk, work, n_pairs,
                          keys=keys, vals=vals, temp=temp, hub_bits=hub_bits)
        if os.environ.get("GN2V_BENCH_MEMLOG"):
            print(f"[mem] prepare: {n_pairs} pairs, room {keys.numel()}, slot {slot}, allocated "
                  f"{torch.cuda.memory_allocated() / 1e9:.1f} GB, reserved "
                  f"{torch.cuda.memory_reserved() / 1e9:.1f} GB", file=sys.stderr, flush=True)
        return keys[:n_pairs], vals[:n_pairs], offsets, n_pairs

    def release(self):
        """Give the standing pair buffers back (before the result tables are assembled)."""
        import torch

        if self._slots or self._temp is not None:
            torch.cuda.synchronize(self.device)
            self._slots, self._temp = {}, None

    def step(self, tp, plan, prepared, alias, cell_rows, central, context, block_id, part, seed,
             epoch, lr, whole_central=False):
        from . import ops

        keys, vals, offsets, n_pairs = prepared
        if n_pairs == 0:
            return
        ops.block_step(self.graph, tp, plan, keys, vals, offsets, alias, cell_rows, central,
                       context, block_id, part, seed, epoch, lr, whole_central=whole_central)


class BlockPartitionedTrainer:
    """See the module docstring.  ``comm``: LoopbackComm / TorchComm (or the threaded stand-in of
    the tests); ``backend``: GpuBlockBackend (tests substitute an oracle-backed one)."""

    def __init__(self, graph, train_params, d: int, ld: int, seed: int, init_scale: float, comm,
                 device, walk_length: int, window: int, min_dist: int = 1,
                 scale_free: bool = True, backend=None, parts: Optional[int] = None,
                 slices: Optional[int] = None, record: int = 16, hot_band=(0, 0),
                 stripes: int = 1):
        """``stripes`` (one GPU only): the centres are split into that many stripes (centre c:
        stripe c % stripes) and a round is trained stripe after stripe, each stripe over the pairs
        of ALL the round's walks whose centre it owns -- what ``stripes`` ranks would do side by
        side.  The pairs of a pass, hence the memory, are those of a round of
        ``walks / stripes`` walks, but a centre's pairs meet in runs ``stripes`` times as long
        (its row is read and added to once per run)."""
        self.graph, self.tp, self.comm = graph, train_params, comm
        self.d, self.ld, self.seed = d, ld, seed
        self.n_nodes = graph.get_number_of_nodes()
        rank, world = comm.rank, comm.world
        self.backend = backend if backend is not None else GpuBlockBackend(graph, device)
        auto_parts, auto_slices = auto_plan(self.n_nodes, world)
        parts = auto_parts if parts is None else parts
        slices = auto_slices if slices is None else slices
        if world > 1 and (parts % world or parts < 2 * world):
            raise ValueError("With several ranks the number of context parts must be a multiple "
                             "of the number of ranks, at least two per rank.")
        self.parts, self.slices = parts, slices
        self.per_rank = parts // world
        self.stripes = max(1, int(stripes))
        if self.stripes > 1 and world > 1:
            raise ValueError("Centre stripes are the one-GPU form of several ranks: stripes > 1 "
                             "needs world == 1.")
        # plans[j]: the view of stripe j (a rank of a world of `stripes`); plan = plans[0]
        self.plans = [
            self.backend.plan(world=self.stripes if self.stripes > 1 else world,
                              rank=j if self.stripes > 1 else rank, parts=parts, slices=slices,
                              walk_length=walk_length, window=window, min_dist=min_dist,
                              record=record, flags=int(train_params.flags) & 2,
                              hot_lo=int(hot_band[0]), hot_hi=int(hot_band[1]))
            for j in range(self.stripes)]
        self.plan = self.plans[0]
        self.scale_free = bool(scale_free)
        # per-cell alias tables for the negatives + the hot-row flags (rows updated by atomics)
        self.alias, self.cell_rows, self.hub_bits = (
            self.backend.alias_tables(self.plan) if scale_free else (None, None, None))
        for p in range(parts):
            if stripe_rows(self.n_nodes, p, parts) == 0:
                raise ValueError("A context part owns no node: graph too small to split this far.")
        # central partition `rank`: never moves
        self.central = self.backend.init_rows(stripe_rows(self.n_nodes, rank, world), d, ld, seed,
                                              0, init_scale, rank, world)
        # context parts held now: {part id: tensor}
        self.max_part_rows = stripe_rows(self.n_nodes, 0, parts)
        self.held = {}
        mine = range(self.per_rank * rank, self.per_rank * (rank + 1))
        for p in mine:
            buf = self.backend.empty_rows(self.max_part_rows, ld)
            rows = stripe_rows(self.n_nodes, p, parts)
            self.backend.init_rows(rows, d, ld, seed, 1, init_scale, p, parts, out=buf[:rows])
            self.held[p] = buf
        self._spare = self.backend.empty_rows(self.max_part_rows, ld) if world > 1 else None
        # walks per rank and round the pair buffers are sized for (None: every round by itself);
        # with it a fit allocates once and a short round reuses the blocks of the full ones
        self.round_capacity = None
        self.episode = 0      # global episode counter g
        self.rounds_done = 0
        self.last_round = None

    # ------------------------------------------------------------------ helpers
    def part_of_episode(self, g: int) -> int:
        world = self.comm.world
        return (self.per_rank * self.comm.rank + g) % self.parts

    def part_rows(self, p: int) -> int:
        return stripe_rows(self.n_nodes, p, self.parts)

    def prepare(self, walks, seed: int, epoch: int, first_walk: int, slot=None, stripe: int = 0):
        """Gather the round's walks from every rank and extract + sort this rank's pairs.
        ``walks``: this rank's int32 [n, L] slice (ids first_walk + rank * n + [0, n)); every rank
        passes the same n (ranks with fewer walks pad with sentinel rows).  ``slot``: which of the
        backend's standing buffers receives the pairs (``run`` alternates two when it overlaps;
        None: buffers of the round's own)."""
        walks_all = self.comm.all_gather(walks)
        if slot is not None and isinstance(self.backend, GpuBlockBackend):
            scale = (self.round_capacity or 0) / max(1, walks.shape[0])
            return self.backend.prepare(self.plans[stripe], walks_all, seed, epoch, first_walk,
                                        self.hub_bits, scale=scale, slot=slot)
        return self.backend.prepare(self.plans[stripe], walks_all, seed, epoch, first_walk,
                                    self.hub_bits)

    def train_prepared(self, prepared, seed: int, epoch: int, lr: float, stripe: int = 0):
        """`parts` episodes over the prepared pairs of one round (of one centre stripe of it)."""
        comm, world = self.comm, self.comm.world
        striped = self.stripes > 1
        block_id = (self.rounds_done * self.stripes + stripe if striped
                    else self.rounds_done * world + comm.rank)
        for _ in range(self.parts):
            g = self.episode
            part = self.part_of_episode(g)
            pending = None
            if world > 1 and g >= 1:
                # the part finished last episode leaves for rank - 1 (which trains it per_rank - 1
                # episodes from now), the part this rank needs per_rank - 1 episodes from now
                # arrives from rank + 1; both while this episode trains
                done = self.part_of_episode(g - 1)
                nxt = self.part_of_episode(g + self.per_rank - 1)
                send_buf = self.held.pop(done)
                recv_buf = self._spare
                pending = comm.sendrecv_start(send_buf[: self.part_rows(done)],
                                              (comm.rank - 1) % world,
                                              recv_buf[: self.part_rows(nxt)],
                                              (comm.rank + 1) % world)
            ctx = self.held[part]
            if striped:
                self.backend.step(self.tp, self.plans[stripe], prepared, self.alias,
                                  self.cell_rows, self.central, ctx[: self.part_rows(part)],
                                  block_id, part, seed, epoch, lr, whole_central=True)
            else:
                self.backend.step(self.tp, self.plan, prepared, self.alias, self.cell_rows,
                                  self.central, ctx[: self.part_rows(part)], block_id, part, seed,
                                  epoch, lr)
            if pending is not None:
                pending.wait()
                self.held[nxt] = recv_buf
                self._spare = send_buf
            self.episode += 1
        trained = int(prepared[3]) + (self.last_round["pairs_trained"] if stripe else 0)
        self.last_round = {"pairs_trained": trained}
        if stripe == self.stripes - 1:
            self.rounds_done += 1

    def train_round(self, walks, seed: int, epoch: int, lr: float, first_walk: int, slot=None):
        for j in range(self.stripes):
            self.train_prepared(self.prepare(walks, seed, epoch, first_walk, slot=slot, stripe=j),
                                seed, epoch, lr, stripe=j)

    def run(self, rounds, overlap: bool = True):
        """Train a sequence of rounds; ``rounds`` is a list of ``(make_walks, seed, epoch, lr,
        first_walk)`` where ``make_walks()`` returns this rank's int32 [n, L] walks of the round
        (called on the stream the preparation runs on).  With ``overlap`` the preparation of round
        t + 1 -- walk generation, all-gather, pair extraction, sort -- runs on a second stream
        while round t trains; at most two rounds are in flight."""
        import torch

        rounds = list(rounds)
        if not rounds:
            return
        on_gpu = isinstance(self.backend, GpuBlockBackend)
        # a single round too: one allocator pool for all rounds; centre stripes run in line
        overlap = overlap and on_gpu and self.stripes == 1
        make, seed, epoch, lr, first = rounds[0]
        if not overlap:
            # one standing slot: the stream orders a round's (a stripe's) training before the
            # pairs of the next are written
            for make, seed, epoch, lr, first in rounds:
                self.train_round(make(), seed, epoch, lr, first, slot=0)
            return
        dev = self.backend.device
        main = torch.cuda.current_stream(dev)
        if getattr(self, "_side", None) is None:
            self._side = torch.cuda.Stream(dev)
        side, before = self._side, None
        # every preparation on the side stream, the first one too: the allocator keeps one pool of
        # freed blocks per stream, and ~100 GB of pair buffers must come back to the pool that the
        # next preparation allocates from
        side.wait_stream(main)
        turn = getattr(self, "_turn", 0)  # two standing slots, alternating across run() calls too
        with torch.cuda.stream(side):
            prepared = self.prepare(make(), seed, epoch, first, slot=turn)
        main.wait_stream(side)
        for t, (_, seed, epoch, lr, _) in enumerate(rounds):
            self.train_prepared(prepared, seed, epoch, lr)
            done = torch.cuda.Event()
            done.record(main)
            nxt = None
            if t + 1 < len(rounds):
                if before is not None:
                    before.synchronize()  # round t - 1 is over: its buffers may be reused
                make, nseed, nepoch, _, nfirst = rounds[t + 1]
                turn ^= 1
                with torch.cuda.stream(side):
                    nxt = self.prepare(make(), nseed, nepoch, nfirst, slot=turn)
                main.wait_stream(side)
            before, prepared = done, nxt
        self._turn = turn ^ 1

    # ------------------------------------------------------------------ results
    def gather_full(self):
        """(central, contextual) as full [N, ld] tables on every rank."""
        import torch

        comm, world, n, ld = self.comm, self.comm.world, self.n_nodes, self.ld
        if hasattr(self.backend, "release"):
            self.backend.release()
        if world == 1:
            # the central partition of the only rank IS the table; parts are released one by one
            if self.parts == 1:
                return self.central, self.held[0]
            context = self.backend.empty_rows(n, ld)
            for p in sorted(self.held):
                context[p::self.parts] = self.held[p][: self.part_rows(p)]
            return self.central, context
        central = self.backend.empty_rows(n, ld)
        context = self.backend.empty_rows(n, ld)
        max_c = stripe_rows(n, 0, world)
        padded = self.backend.empty_rows(max_c, ld)
        padded.zero_()
        padded[: self.central.shape[0]] = self.central
        every = comm.all_gather(padded)
        for r in range(world):
            central[r::world] = every[r * max_c: r * max_c + stripe_rows(n, r, world)]
        del every, padded
        ids = sorted(self.held)
        mine = torch.stack([self.held[p] for p in ids])          # [2, max_rows, ld]
        id_t = torch.tensor(ids, dtype=torch.int64, device=mine.device)
        all_ids = comm.all_gather(id_t).tolist()
        every = comm.all_gather(mine)
        for i, p in enumerate(all_ids):
            context[p::self.parts] = every[i][: self.part_rows(p)]
        return central, context

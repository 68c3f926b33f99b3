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
  centre it owns -- one 64-bit word per pair, a *group* of consecutive episodes' parts at a time:
  two passes over the walks and one radix sort in HIP (``gn2v_block_count`` /
  ``gn2v_block_extract``), so the pair buffers hold 1 / groups of the round; nothing else
  crosses the fabric.
* Training: ``gn2v_block_step`` per part, negatives drawn degree-proportionally inside the
  cell of the pair's context (one alias table per cell).
* Placement (resident cells, i.e. plans of more than 16 slices): which cell a node's contextual
  row is trained in changes every round -- ``gn2v_block_placement``, a seeded permutation of the
  node ids inside their classes modulo ``parts`` (several ranks: a row never leaves its part) or
  over the whole graph (one GPU: the table stays in node order and the kernel reaches the rows
  through the placement's inverse) -- so that the negatives a context meets over a fit range
  over the part / the graph, not over one fixed set of ~200 cell-mates
  (node2vec_skipgram.py:101-102).  The alias tables follow the placement, round by round.

Preparation of the next group (for the first group of a round: walk generation and all-gather;
then extraction and sort) runs on a second stream while the current group trains.
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


def rounds_per_epoch(epochs: int) -> int:
    """Rounds an epoch of the graph is cut into under a placement (csrc/handle.h
    ``rounds_per_epoch``: 192 over the fit, 16 to 64 an epoch; GN2V_ROUNDS_PER_EPOCH pins it)."""
    pinned = os.environ.get("GN2V_ROUNDS_PER_EPOCH", "")
    if pinned:
        return max(1, int(pinned))
    e = max(1, int(epochs))
    return min(64, max(16, -(-192 // e)))


PAIR_ROOM = 1 << 24  # pair buffers are sized in steps of this many pairs
MIN_ROWS_PER_CELL = 32768


def round_plan(free_bytes: int, n_nodes: int, walk_length: int, window: int, world: int,
               parts: int, slices: int, overlap: bool, cap: int = 0) -> Tuple[int, int]:
    """(walks per rank and round, parts per extraction group) for ``free_bytes`` of HBM
    (``gn2v_block_round_plan``: a round long enough for 64 pairs per (cell, centre) within
    [2^20, 2^23] walks, at most ``cap`` when given -- the rounds-per-epoch rule, a caller's round:
    the groups are sized for the round that will be trained; equal groups of parts -- at least
    four per round (resident cells: ONE when memory allows, on one GPU and with several ranks;
    include/gn2v_internal.h) -- whose pair words, held once sorted (twice when the next group is
    prepared meanwhile) and once unsorted, fit three quarters of it beside the walks).  Every rank
    must use the same values."""
    import ctypes as C

    from . import _lib

    walks, group = C.c_uint64(max(0, int(cap))), C.c_uint32()
    _lib.check(_lib.lib().gn2v_block_round_plan(int(free_bytes), n_nodes, walk_length, window,
                                                world, parts, slices, int(bool(overlap)),
                                                C.byref(walks), C.byref(group)))
    return walks.value, group.value


def auto_plan(n_nodes: int, world: int, ld: int = 0, k: int = 10) -> Tuple[int, int]:
    """(parts, slices) of the contextual table (``gn2v_block_auto_plan``: one rule for the C++
    one-GPU fit and for this trainer).  Row stride ``ld`` <= 512 floats (0: unknown),
    up to 115 M nodes: resident cells -- cells that fit one workgroup's LDS, whose rows are
    read and updated there by that workgroup alone (plain read-modify-writes of its sixteen
    waves: no other CU races for a row, the workgroup's own groups still can).  Otherwise XCD
    cells.  Measured (scripts/quality_probe.py):
    the smaller a cell, the more of it lives in the XCD's L2 (BA 10 M nodes: 0.67 of the HBM
    roofline unsliced, 0.80 at 4 x 8 cells, 0.87 at 32 x 8) -- and the more often two waves
    read-modify-write the same row at once; link quality stays at or above the walk-ordered
    trainer's while a cell keeps >= 32 k rows.  slices = 8 (one per XCD: only then is a row
    exclusive to one L2; graphs too small for cells of 8 k rows: 1); as many parts -- any count,
    a multiple of the ranks, at least two per rank -- as keep MIN_ROWS_PER_CELL rows a cell."""
    import ctypes as C

    from . import _lib

    parts, slices = C.c_uint32(), C.c_uint32()
    _lib.check(_lib.lib().gn2v_block_auto_plan(n_nodes, world, int(ld), int(k), C.byref(parts),
                                               C.byref(slices)))
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

    def broadcast(self, tensor, src):
        return tensor

    def send_to_root(self, tensor, root, buffer=None, src=None):
        return tensor


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

    def broadcast(self, tensor, src):
        """`tensor` of rank `src` into `tensor` of every rank (in place)."""
        if self._staged and tensor.is_cuda:
            host = tensor.cpu()
            self._dist.broadcast(host, src, group=self._group)
            tensor.copy_(host)
            return tensor
        self._dist.broadcast(tensor, src, group=self._group)
        return tensor

    def send_to_root(self, tensor, root, buffer=None, src=None):
        """Rank `src` sends `tensor`; rank `root` receives it into `buffer` (same shape) and gets
        it back; every other rank gets None.  src == root: no transfer."""
        dist = self._dist
        if src == root:
            return tensor if self.rank == root else None
        if self.rank == src:
            t = tensor.cpu() if self._staged and tensor.is_cuda else tensor
            dist.send(t.contiguous(), root, group=self._group)
            return None
        if self.rank == root:
            if self._staged and buffer.is_cuda:
                host = buffer.cpu()
                dist.recv(host, src, group=self._group)
                buffer.copy_(host)
            else:
                dist.recv(buffer, src, group=self._group)
            return buffer
        return None


class _DeviceBytes:
    """A raw device range as an object torch.as_tensor understands (CUDA array interface)."""

    def __init__(self, ptr, nbytes):
        self.__cuda_array_interface__ = {"shape": (int(nbytes),), "typestr": "|u1",
                                         "data": (int(ptr), False), "version": 2}


class CComm:
    """``gn2v_comm`` (include/gn2v.h) over one of this module's communicators: what the host hands
    to ``gn2v_train_world`` so that the C loop moves its walks and parts through
    torch.distributed (RCCL: ``TorchComm``), through the threads of a test, or not at all
    (``LoopbackComm``).  The callbacks see raw device pointers and byte counts and wrap them as
    uint8 tensors without a copy; they run on the calling thread with the GIL held and make the
    stream they are handed torch's current one.  Keep the
    object alive for as long as the C call runs (it owns the ctypes callbacks)."""

    def __init__(self, comm, device):
        import torch

        from . import _lib

        self.comm, self.device = comm, torch.device(device)
        self._pending, self._next, self.error = {}, 1, None

        def view(ptr, nbytes):
            return torch.as_tensor(_DeviceBytes(ptr, nbytes), device=self.device)

        def guarded(fn, stream_arg):
            def call(*args):
                try:
                    with torch.cuda.device(self.device):
                        # the caller's stream (args[stream_arg], a hipStream_t) becomes torch's
                        # current one for the call: the communicators order their work after it
                        raw = args[stream_arg]
                        if raw:
                            with torch.cuda.stream(torch.cuda.ExternalStream(int(raw))):
                                fn(*args)
                        else:
                            fn(*args)
                    return 0
                except Exception as e:  # noqa: BLE001 -- a Python error must not cross the C frame
                    self.error = e
                    return 1
            return call

        def all_gather(_ctx, send, recv, nbytes, _stream):
            out = comm.all_gather(view(send, nbytes))
            view(recv, nbytes * comm.world).copy_(out.reshape(-1))

        def sendrecv_start(_ctx, send, send_bytes, dst, recv, recv_bytes, src, _stream, handle):
            pending = comm.sendrecv_start(view(send, send_bytes), int(dst), view(recv, recv_bytes),
                                          int(src))
            key, self._next = self._next, self._next + 1
            self._pending[key] = pending
            handle[0] = key

        def sendrecv_wait(_ctx, handle, _stream):
            self._pending.pop(int(handle)).wait()

        def broadcast(_ctx, buf, nbytes, root, _stream):
            comm.broadcast(view(buf, nbytes), int(root))

        self._callbacks = (_lib.COMM_ALL_GATHER(guarded(all_gather, 4)),
                           _lib.COMM_SENDRECV_START(guarded(sendrecv_start, 7)),
                           _lib.COMM_SENDRECV_WAIT(guarded(sendrecv_wait, 2)),
                           _lib.COMM_BROADCAST(guarded(broadcast, 4)))
        self.struct = _lib.Comm(None, comm.rank, comm.world, *self._callbacks)


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

    def auto_plan(self, world, ld, k):
        """(parts, slices) by ``gn2v_block_auto_plan_graph``: the rule of ``auto_plan`` with the
        graph at hand (resident cells only without a hub that would hold every launch up)."""
        import ctypes as C

        import torch

        from . import _lib

        parts, slices = C.c_uint32(), C.c_uint32()
        dg = self.graph.device_graph(self.index)
        with torch.cuda.device(self.device):
            stream = torch.cuda.current_stream().cuda_stream
            _lib.check(_lib.lib().gn2v_block_auto_plan_graph(
                dg.handle, world, int(ld), int(k), C.byref(parts), C.byref(slices), stream))
        return parts.value, slices.value

    def alias_tables(self, plan, inv=None, out=None):
        from . import ops

        return ops.block_alias(self.graph, plan, device=self.index, inv=inv, out=out)

    def placement(self, classes, seed, round_id, out=None):
        """(place, inv) of the round (``gn2v_block_placement``)."""
        from . import ops

        return ops.block_placement(self.graph, classes, seed, round_id, device=self.index, out=out)

    def place_walks(self, place, walks_all):
        from . import ops

        return ops.block_place_walks(place, walks_all)

    def prepare(self, plan, walks_all, seed, epoch, first_walk, hub_bits=None, part_lo=0,
                part_n=0, capacity=0, slot=None, placed=None):
        """-> (pairs, cell_offsets, n_pairs): the sorted pair words of the group of parts
        ``part_lo, part_lo + 1, ...`` (``part_n`` of them, cyclic; 0, 0 = every part); one host
        read (the pair count).

        ``slot`` None: fresh buffers of this group's size.  ``slot`` 0 / 1: the trainer's standing
        buffers -- one pair buffer per slot and one buffer for the unsorted words + the sort,
        allocated once for ``capacity`` pairs (what a full group of a full round holds; a short
        first or last round reuses them) and kept until ``release()``: tens of GB must not depend
        on which cached block the allocator happens to split."""
        from . import ops

        import torch

        work, offsets = ops.block_count(self.graph, plan, walks_all, seed, epoch, first_walk,
                                        part_lo=part_lo, part_n=part_n, placed=placed)
        n_pairs = int(offsets[-1])
        dev = walks_all.device

        def rounded(n):  # in steps of 2^24 pairs
            return max(1, -(-int(n) // PAIR_ROOM)) * PAIR_ROOM

        if slot is None:
            room = rounded(n_pairs)
            pairs = torch.empty(room, dtype=torch.int64, device=dev)
            temp = torch.empty(ops.block_extract_temp_bytes(room), dtype=torch.uint8, device=dev)
        else:
            pairs = self._slots.get(slot)
            if pairs is None or pairs.numel() < n_pairs:
                self._slots[slot] = pairs = None  # released before its successor is allocated
                # the groups of a fit differ by a few percent: 1/64 on top of the largest seen
                room = rounded(max(n_pairs, capacity) * (1 + 1 / 64))
                pairs = self._slots[slot] = torch.empty(room, dtype=torch.int64, device=dev)
            need = ops.block_extract_temp_bytes(pairs.numel())
            if self._temp is None or self._temp.numel() < need:
                self._temp = None
                self._temp = torch.empty(need, dtype=torch.uint8, device=dev)
            temp = self._temp
        ops.block_extract(self.graph, plan, walks_all, seed, epoch, first_walk, work, n_pairs,
                          pairs=pairs, temp=temp, hub_bits=hub_bits, part_lo=part_lo,
                          part_n=part_n, placed=placed)
        # (a wide group -- more cells than the counting pass counts in LDS -- gets its cell
        # offsets from the sorted words; a counted group has them already)
        ops.block_cell_offsets(self.graph, plan, part_n, pairs, n_pairs, offsets)
        if os.environ.get("GN2V_BENCH_MEMLOG"):
            print(f"[mem] prepare: parts {part_lo}+{part_n}: {n_pairs} pairs, room "
                  f"{pairs.numel()}, slot {slot}, allocated "
                  f"{torch.cuda.memory_allocated() / 1e9:.1f} GB, reserved "
                  f"{torch.cuda.memory_reserved() / 1e9:.1f} GB", file=sys.stderr, flush=True)
        return pairs[:n_pairs], offsets, n_pairs

    def round(self, tp, plans, walks, tables, central, part_rows, group_parts, capacity, seed,
              epoch, first_walk, lr, round_id, placed=None, inv=None, context_table=None):
        """A whole round on one GPU through the C round driver (``gn2v_block_round``: the loop
        ``gn2v_train_blocks`` runs too -- one host loop orders the launches of a one-GPU fit):
        every stripe in ``plans``, every group of ``group_parts`` parts: count, extract + sort,
        one step per part.  ``tables``: (alias, cell_rows, hub_bits, hot_list, hot_slot) or
        Nones; ``central``: the whole table; ``part_rows[p]``: the rows of part p -- or, under
        a placement (``placed``: the walks with placed ids, ``inv``), ``context_table``: the whole
        contextual table in node order.  Pairs in the standing buffers of slot 0 (``capacity``
        pairs to begin with, grown when the driver asks for it).  Returns the pairs trained."""
        import ctypes as C

        import torch

        from . import _lib, ops

        dev = walks.device
        plan = plans[0]
        ptr = lambda t: None if t is None else t.data_ptr()  # noqa: E731
        if getattr(self, "_work", None) is None:
            self._work = torch.empty(_lib.BLOCK_WORK_WORDS, dtype=torch.int64, device=dev)
        cells = plan.parts * plan.slices
        if getattr(self, "_offsets", None) is None or self._offsets.numel() != cells + 1:
            self._offsets = torch.empty(cells + 1, dtype=torch.int64, device=dev)

        def rounded(n):  # in steps of 2^24 pairs
            return max(1, -(-int(n) // PAIR_ROOM)) * PAIR_ROOM

        def room_for(n_pairs):
            pairs = self._slots.get(0)
            if pairs is None or pairs.numel() < n_pairs:
                self._slots[0] = pairs = None  # released before its successor is allocated
                pairs = self._slots[0] = torch.empty(rounded(n_pairs * (1 + 1 / 64)),
                                                     dtype=torch.int64, device=dev)
            need = ops.block_extract_temp_bytes(pairs.numel())
            if self._temp is None or self._temp.numel() < need:
                self._temp = None
                self._temp = torch.empty(need, dtype=torch.uint8, device=dev)
            return pairs, self._temp

        pairs, temp = room_for(max(1, capacity))
        alias, cell_rows, hub_bits, hot_list, hot_slot = tables
        assert walks.is_contiguous() and central.is_contiguous()
        context_ld = 0
        if context_table is not None:  # parts = the rows p, p + parts, ... of the one table
            assert context_table.is_contiguous() and context_table.shape[1] == tp.ld
            part_ptrs = (C.c_void_p * plan.parts)(
                *[context_table.data_ptr() + p * tp.ld * 4 for p in range(plan.parts)])
            context_ld = plan.parts * tp.ld
        else:
            assert all(t.is_contiguous() and t.shape[1] == tp.ld for t in part_rows)
            part_ptrs = (C.c_void_p * len(part_rows))(*[t.data_ptr() for t in part_rows])
        plan_array = (_lib.BlockPlan * len(plans))(*plans)
        io = _lib.BlockRoundIO(
            ptr(walks), ptr(placed), ptr(inv), ptr(context_table),
            ptr(alias), ptr(cell_rows), ptr(hub_bits), ptr(hot_list), ptr(hot_slot),
            ptr(central), C.cast(part_ptrs, C.POINTER(C.c_void_p)), context_ld, ptr(self._work),
            ptr(self._offsets), ptr(pairs), pairs.numel(), ptr(temp), temp.numel(),
            int(group_parts), 0, 0, 0)
        dg = self.graph.device_graph(self.index)
        while True:
            rc = _lib.lib().gn2v_block_round(
                dg.handle, C.byref(tp), plan_array, len(plans), C.byref(io), walks.shape[0],
                seed, epoch, first_walk, lr, round_id, ops._stream(dev))
            if rc != _lib.ROUND_GROW:
                _lib.check(rc)
                return int(io.pairs_trained)
            pairs, temp = room_for(int(io.needed_pairs))
            io.d_pairs, io.pairs_capacity = ptr(pairs), pairs.numel()
            io.d_temp, io.temp_bytes = ptr(temp), temp.numel()

    def release(self):
        """Give the standing pair buffers back (before the result tables are assembled)."""
        import torch

        if self._slots or self._temp is not None:
            torch.cuda.synchronize(self.device)
            self._slots, self._temp = {}, None

    def step(self, tp, plan, prepared, alias, cell_rows, central, context, block_id, part, seed,
             epoch, lr, whole_central=False, hot=None, inv=None, context_table=None):
        from . import ops

        pairs, offsets, n_pairs = prepared[:3]
        if n_pairs == 0:
            return
        ops.block_step(self.graph, tp, plan, pairs, offsets, alias, cell_rows, central,
                       context, block_id, part, seed, epoch, lr, whole_central=whole_central,
                       hot=hot, inv=inv, context_table=context_table)


class BlockPartitionedTrainer:
    """See the module docstring.  ``comm``: LoopbackComm / TorchComm (or the threaded stand-in of
    the tests); ``backend``: GpuBlockBackend (tests substitute an oracle-backed one)."""

    def __init__(self, graph, train_params, d: int, ld: int, seed: int, init_scale: float, comm,
                 device, walk_length: int, window: int, min_dist: int = 1,
                 scale_free: bool = True, backend=None, parts: Optional[int] = None,
                 slices: Optional[int] = None, record: int = 32, hot_rows: Optional[int] = None,
                 hot_flush: int = 0, stripes: int = 1, group_parts: Optional[int] = None):
        """``group_parts``: the parts whose pairs are extracted, sorted and held at a time (None:
        all of a round at once; ``models.fit_transform_blocks`` and ``bench.py`` take it from
        ``round_plan``).  Every rank must pass the same value.

        ``hot_rows``: the rows of every cell with the highest in-degrees whose updates are
        accumulated in LDS and handed to the row with atomics (None: ``BLOCK_HOT_DEFAULT``; 0:
        none, every row takes the plain stores); ``hot_flush``: see ``ops.block_plan``.

        ``stripes`` (one GPU only): the centres are split into that many stripes (centre c:
        stripe c % stripes) and a round is trained stripe after stripe, each stripe over the pairs
        of ALL the round's walks whose centre it owns -- what ``stripes`` ranks would do side by
        side.  The pairs of a pass, hence the memory, are those of a round of
        ``walks / stripes`` walks, but a centre's pairs meet in runs ``stripes`` times as long
        (its row is read and added to once per run)."""
        self.graph, self.tp, self.comm = graph, train_params, comm
        self.d, self.ld, self.seed = d, ld, seed
        self.n_nodes = graph.get_number_of_nodes()
        self.walk_length, self.window = walk_length, window
        rank, world = comm.rank, comm.world
        self.backend = backend if backend is not None else GpuBlockBackend(graph, device)
        # one GPU: a round is one call of the library's round driver (gn2v_block_round, what
        # gn2v_train_blocks runs); the per-group loop below, with its hops and its overlapped
        # preparation, is for several ranks (tests switch this off to run that loop on one GPU)
        self.round_driver = world == 1 and hasattr(self.backend, "round")
        auto_parts, auto_slices = (
            self.backend.auto_plan(world, ld, int(train_params.k))
            if hasattr(self.backend, "auto_plan")
            else auto_plan(self.n_nodes, world, ld, int(train_params.k)))
        parts = auto_parts if parts is None else parts
        slices = auto_slices if slices is None else slices
        if world > 1 and (parts % world or parts < 2 * world):
            raise ValueError("With several ranks the number of context parts must be a multiple "
                             "of the number of ranks, at least two per rank.")
        self.parts, self.slices = parts, slices
        self.per_rank = parts // world
        # Resident cells (more than 16 slices; gn2v_block_step's rule): no hot rows -- every row
        # of a cell lives in LDS -- and a PLACEMENT per round (module docstring): inside the
        # classes modulo `parts` when the parts travel, over the whole graph on one GPU, where
        # the contextual table then stays ONE table in node order.  GN2V_BLOCK_PERMUTE=0: the
        # fixed cells of round 4 (A/B; every rank alike).
        resident_plan = slices > 16
        self.permute = (resident_plan and os.environ.get("GN2V_BLOCK_PERMUTE", "1") != "0"
                        and hasattr(self.backend, "placement"))
        self.classes = parts if world > 1 else 1
        self.natural = self.permute and world == 1
        if hot_rows is None:
            from . import _lib

            hot_rows = 0 if resident_plan else _lib.BLOCK_HOT_DEFAULT
        if self.permute:
            hot_rows = 0
        # the extraction counts the cells of a group in LDS: BLOCK_MAX_GROUP_CELLS at most
        from . import _lib as _l

        # (and fewer when the walk's staging leaves less of the 64 KB: gn2v_block_round_plan)
        staging = 4 * (5 if walk_length <= 128 else 4) * walk_length * 4
        most = max(1, min(_l.BLOCK_MAX_GROUP_CELLS, (64 * 1024 - min(staging, 60 * 1024)) // 4)
                   // slices)
        if world > 1 and slices > 16:  # wide groups: their offsets follow the sort
            most = max(1, _l.BLOCK_MAX_WIDE_GROUP_CELLS // slices)
        self.group_parts = min(most, parts if not group_parts else max(1, min(int(group_parts), parts)))
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
                              hot_rows=int(hot_rows), hot_flush=int(hot_flush))
            for j in range(self.stripes)]
        self.plan = self.plans[0]
        self.scale_free = bool(scale_free)
        # per-cell alias tables for the negatives + the hot rows of every cell (flags, slots)
        # (with uniform negatives the alias tables are not used: the hot rows, the most frequent
        # contexts, still are)
        # (under a placement the tables follow it round by round: _round_state)
        self.alias, self.cell_rows, self.hub_bits, hot_list, hot_slot = (
            self.backend.alias_tables(self.plan)
            if (scale_free or hot_rows) and not self.permute else (None,) * 5)
        self.hot = (hot_list, hot_slot) if hot_list is not None and hot_rows else None
        if not scale_free:
            self.alias = None
        self._rstates = [None, None]  # placement, alias tables, placed walks of two rounds
        for p in range(parts):
            if stripe_rows(self.n_nodes, p, parts) == 0:
                raise ValueError("A context part owns no node: graph too small to split this far.")
        # central partition `rank`: never moves
        self.central = self.backend.init_rows(stripe_rows(self.n_nodes, rank, world), d, ld, seed,
                                              0, init_scale, rank, world)
        # context parts held now: {part id: tensor}
        self.max_part_rows = stripe_rows(self.n_nodes, 0, parts)
        self.held = {}
        self.context_table = None
        mine = range(self.per_rank * rank, self.per_rank * (rank + 1))
        if self.natural:
            # one table in node order; part p = its rows p, p + parts, ... (views)
            self.context_table = self.backend.init_rows(self.n_nodes, d, ld, seed, 1, init_scale,
                                                        0, 1)
            for p in mine:
                self.held[p] = self.context_table[p::parts]
            mine = ()
        for p in mine:
            buf = self.backend.empty_rows(self.max_part_rows, ld)
            rows = stripe_rows(self.n_nodes, p, parts)
            self.backend.init_rows(rows, d, ld, seed, 1, init_scale, p, parts, out=buf[:rows])
            self.held[p] = buf
        self._spare = self.backend.empty_rows(self.max_part_rows, ld) if world > 1 else None
        # walks per rank and round the pair buffers are sized for (None: every group by itself);
        # with it a fit allocates once and a short round reuses the blocks of the full ones
        self.round_capacity = None
        self.episode = 0      # global episode counter g
        self.rounds_done = 0
        self._round_episodes = 0  # episodes of the current round so far (all stripes)
        self.last_round = None
        self.wait_ms = []     # HIP events around the wait for each hop (timed=True in run())
        self.spans = []       # (phase, start event, end event) of run(timed=True): phase_ms()

    # ------------------------------------------------------------------ helpers
    def part_of_episode(self, g: int) -> int:
        return (self.per_rank * self.comm.rank + g) % self.parts

    def part_rows(self, p: int) -> int:
        return stripe_rows(self.n_nodes, p, self.parts)

    def groups(self) -> List[Tuple[int, int]]:
        """(first part, number of parts) of the groups a round is prepared and trained in, in
        this rank's episode order (a round starts at an episode that is a multiple of `parts`:
        this rank's first part is always per_rank * rank; the parts of a group are consecutive
        modulo `parts`)."""
        gp = self.group_parts
        return [(self.part_of_episode(e), min(gp, self.parts - e))
                for e in range(0, self.parts, gp)]

    def group_capacity(self) -> int:
        """Pairs the standing buffers are sized for: a full group of a full round (untrimmed
        windows) plus 1/8 -- parts, and with them groups, are not equally heavy."""
        if not self.round_capacity:
            return 0
        pairs = self.round_capacity * 2 * self.window * self.walk_length
        share = pairs * min(self.group_parts, self.parts) // self.parts
        return share + share // 8

    def gather_walks(self, walks):
        """This rank's int32 [n, L] walks of the round -> the walks of every rank (all-gathered;
        ids first_walk + rank * n + [0, n)); every rank passes the same n (ranks with fewer
        walks pad with sentinel rows)."""
        return self.comm.all_gather(walks)

    def round_state(self, walks_all, seed: int, round_id: int):
        """What a round under a placement needs beside its walks: the placement itself, the
        alias tables that follow it and the walks with placed ids (None without a placement).
        Two sets alternate, so that the preparation of round r + 1 never touches what the
        training of round r still reads."""
        if not self.permute:
            return None
        turn = round_id & 1
        st = self._rstates[turn]
        if st is None:
            st = self._rstates[turn] = {"placement": None, "alias": None}
        st["placement"] = self.backend.placement(self.classes, seed, round_id,
                                                 out=st["placement"])
        place, inv = st["placement"]
        if self.scale_free:
            tables = self.backend.alias_tables(self.plan, inv=inv, out=st["alias"])
            st["alias"] = tables[:2]
        st["placed"] = self.backend.place_walks(place, walks_all)
        st["round_id"] = round_id
        return st

    def prepare(self, walks_all, seed: int, epoch: int, first_walk: int, group=None, slot=None,
                stripe: int = 0, rstate=None):
        """Extract + sort this rank's pairs of one group of parts (``group`` = (first part,
        number of parts), None = the whole round) from the round's gathered walks.  ``slot``:
        which of the backend's standing buffers receives the pairs (``run`` alternates two when it
        overlaps; None: buffers of the group's own)."""
        lo, n = group if group is not None else (self.part_of_episode(0), self.parts)
        kw = {}
        if slot is not None and isinstance(self.backend, GpuBlockBackend):
            kw = {"capacity": self.group_capacity(), "slot": slot}
        if rstate is not None:
            kw["placed"] = rstate["placed"]
        pairs, offsets, n_pairs = self.backend.prepare(
            self.plans[stripe], walks_all, seed, epoch, first_walk, self.hub_bits, part_lo=lo,
            part_n=n, **kw)
        return pairs, offsets, n_pairs, lo, n, rstate

    def train_prepared(self, prepared, seed: int, epoch: int, lr: float, stripe: int = 0,
                       timed: bool = False):
        """The episodes of one prepared group (all `parts` of them when the group is a round)."""
        comm, world = self.comm, self.comm.world
        striped = self.stripes > 1
        block_id = (self.rounds_done * self.stripes + stripe if striped
                    else self.rounds_done * world + comm.rank)
        lo, n_parts = prepared[3], prepared[4]
        assert self.part_of_episode(self.episode) == lo, "groups must be trained in episode order"
        for _ in range(n_parts):
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
            rstate = prepared[5] if len(prepared) > 5 else None
            if rstate is None:
                self.backend.step(self.tp, self.plans[stripe], prepared, self.alias,
                                  self.cell_rows, self.central,
                                  self.held[part][: self.part_rows(part)], block_id, part, seed,
                                  epoch, lr, whole_central=striped, hot=self.hot)
            else:
                alias, cell_rows = rstate["alias"] if rstate["alias"] is not None else (None, None)
                self.backend.step(self.tp, self.plans[stripe], prepared, alias, cell_rows,
                                  self.central,
                                  None if self.natural else self.held[part][: self.part_rows(part)],
                                  block_id, part, seed, epoch, lr, whole_central=striped,
                                  inv=rstate["placement"][1], context_table=self.context_table)
            if pending is not None:
                if timed:
                    self._timed_wait(pending)
                else:
                    pending.wait()
                self.held[nxt] = recv_buf
                self._spare = send_buf
            self.episode += 1
        if self._round_episodes == 0:
            self.last_round = {"pairs_trained": 0}
        self.last_round["pairs_trained"] += int(prepared[2])
        self._round_episodes += n_parts
        if self._round_episodes == self.parts * self.stripes:
            self._round_episodes = 0
            self.rounds_done += 1

    def _timed_wait(self, pending):
        """pending.wait() between two events on the current stream: what the compute stream
        really waits for a hop that had a whole episode to complete (read with hop_wait_ms())."""
        import torch

        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        pending.wait()
        b.record()
        self.wait_ms.append((a, b))

    def _span(self, name, stream=None):
        """Context manager: HIP events on ``stream`` (default: the current one) around a phase of
        run(timed=True); read with phase_ms()."""
        import contextlib

        import torch

        trainer = self

        @contextlib.contextmanager
        def span():
            s = stream if stream is not None else torch.cuda.current_stream()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(s)
            try:
                yield
            finally:
                b.record(s)
                trainer.spans.append((name, a, b))

        return span()

    def phase_ms(self):
        """{phase: total ms} of the spans recorded since the last call (after a synchronize):
        walk generation, walk all-gather, pair extraction + sort (all on the preparation stream),
        training, and what the compute stream waited for the preparation ("exposed")."""
        out = {}
        for name, a, b in self.spans:
            out[name] = out.get(name, 0.0) + a.elapsed_time(b)
        self.spans = []
        return out

    def hop_wait_ms(self):
        """Exposed wait per hop in ms (after a synchronize): list of floats."""
        out = [a.elapsed_time(b) for a, b in self.wait_ms]
        self.wait_ms = []
        return out

    def train_round(self, walks, seed: int, epoch: int, lr: float, first_walk: int, slot=None):
        if self.round_driver:
            # one GPU: the C round driver (what gn2v_train_blocks runs) orders the launches
            assert self._round_episodes == 0, "a round driven in one piece starts at its beginning"
            hot_list, hot_slot = self.hot if self.hot is not None else (None, None)
            tables = (self.alias, self.cell_rows, self.hub_bits, hot_list, hot_slot)
            kw = {}
            rstate = self.round_state(walks, seed, self.rounds_done)
            if rstate is not None:
                alias, cell_rows = rstate["alias"] if rstate["alias"] is not None else (None, None)
                tables = (alias, cell_rows, None, None, None)
                kw = {"placed": rstate["placed"], "inv": rstate["placement"][1],
                      "context_table": self.context_table}
            part_rows = (None if self.natural
                         else [self.held[p][: self.part_rows(p)] for p in range(self.parts)])
            trained = self.backend.round(
                self.tp, self.plans, walks, tables, self.central, part_rows, self.group_parts,
                self.group_capacity(), seed, epoch, lr=lr, first_walk=first_walk,
                round_id=self.rounds_done, **kw)
            if rstate is not None:
                rstate["placed"] = None
            self.last_round = {"pairs_trained": trained}
            self.episode += self.parts * self.stripes
            self.rounds_done += 1
            return
        walks_all = self.gather_walks(walks)
        rstate = self.round_state(walks_all, seed, self.rounds_done)
        for j in range(self.stripes):
            for group in self.groups():
                self.train_prepared(
                    self.prepare(walks_all, seed, epoch, first_walk, group=group, slot=slot,
                                 stripe=j, rstate=rstate), seed, epoch, lr, stripe=j)
        if rstate is not None:
            rstate["placed"] = None

    def run(self, rounds, overlap: bool = True, timed: bool = False):
        """Train a sequence of rounds; ``rounds`` is a list of ``(make_walks, seed, epoch, lr,
        first_walk)`` where ``make_walks()`` returns this rank's int32 [n, L] walks of the round
        (called on the stream the preparation runs on).  With ``overlap`` the preparation of the
        next group -- for the first group of a round: walk generation and all-gather; then pair
        extraction and sort -- runs on a second stream while the current group trains; two groups
        are in flight at most.  ``timed``: HIP events around every wait for a hop
        (``hop_wait_ms()``)."""
        import torch

        rounds = list(rounds)
        if not rounds:
            return
        on_gpu = isinstance(self.backend, GpuBlockBackend)
        # a single round too: one allocator pool for all rounds; centre stripes run in line, and
        # so does one GPU by itself (train_round: the C round driver; the kernels of a part take
        # 96 % of a round there, and the preparation has no exchange to hide)
        overlap = overlap and on_gpu and self.stripes == 1 and not self.round_driver
        if not overlap:
            # one standing slot: the stream orders a group's training before the pairs of the
            # next are written
            for make, seed, epoch, lr, first in rounds:
                self.train_round(make(), seed, epoch, lr, first, slot=0 if on_gpu else None)
            return
        dev = self.backend.device
        main = torch.cuda.current_stream(dev)
        if getattr(self, "_side", None) is None:
            self._side = torch.cuda.Stream(dev)
        side = self._side
        groups = self.groups()
        units = [(r, gi) for r in range(len(rounds)) for gi in range(len(groups))]
        state = {"walks_all": None, "rstate": None}
        assert self._round_episodes == 0, "run() starts at the beginning of a round"
        first_round = self.rounds_done

        def prep(unit, turn):
            r, gi = unit
            make, seed, epoch, _, first = rounds[r]
            # every preparation on the side stream, the first one too: the allocator keeps one
            # pool of freed blocks per stream, and tens of GB of walks and pair buffers must come
            # back to the pool that the next preparation allocates from
            with torch.cuda.stream(side):
                if gi == 0:
                    state["walks_all"] = None  # the last round's walks go back to the pool first
                    if timed:
                        with self._span("walk_generation"):
                            mine = make()
                        with self._span("walk_allgather"):
                            state["walks_all"] = self.gather_walks(mine)
                        del mine
                    else:
                        state["walks_all"] = self.gather_walks(make())
                    # the round's placement, alias tables and placed walks (resident cells)
                    if state["rstate"] is not None:
                        state["rstate"]["placed"] = None
                    state["rstate"] = self.round_state(state["walks_all"], seed, first_round + r)
                if timed:
                    with self._span("extract_and_sort"):
                        return self.prepare(state["walks_all"], seed, epoch, first,
                                            group=groups[gi], slot=turn, rstate=state["rstate"])
                return self.prepare(state["walks_all"], seed, epoch, first, group=groups[gi],
                                    slot=turn, rstate=state["rstate"])

        def wait_for_preparation():
            # what the compute stream really waits for the preparation stream: exposed time
            if timed:
                with self._span("exposed_preparation_wait", main):
                    main.wait_stream(side)
            else:
                main.wait_stream(side)

        side.wait_stream(main)
        turn = getattr(self, "_turn", 0)  # two standing slots, alternating across run() calls too
        prepared = prep(units[0], turn)
        wait_for_preparation()
        before = None
        for u, (r, gi) in enumerate(units):
            _, seed, epoch, lr, _ = rounds[r]
            if timed:
                with self._span("training", main):
                    self.train_prepared(prepared, seed, epoch, lr, timed=timed)
            else:
                self.train_prepared(prepared, seed, epoch, lr, timed=timed)
            done = torch.cuda.Event()
            done.record(main)
            nxt = None
            if u + 1 < len(units):
                if before is not None:
                    side.wait_event(before)  # group u - 1 is over: its slot may be rewritten
                turn ^= 1
                nxt = prep(units[u + 1], turn)
                wait_for_preparation()
            before, prepared = done, nxt
        state["walks_all"] = None
        if state["rstate"] is not None:
            state["rstate"]["placed"] = None
        self._turn = turn ^ 1

    # ------------------------------------------------------------------ results
    def gather_full(self, root: Optional[int] = None):
        """(central, contextual) as full [N, ld] tables: on every rank (``root`` None), or on rank
        ``root`` only (the other ranks get ``(None, None)`` and hold nothing beyond their own
        shards).  Partitions travel one at a time through a staging buffer of one partition: a
        rank that assembles holds the two tables plus that buffer, never a second copy."""
        comm, world, n, ld = self.comm, self.comm.world, self.n_nodes, self.ld
        if hasattr(self.backend, "release"):
            self.backend.release()
        if world == 1:
            # the central partition of the only rank IS the table; parts are released one by one
            if self.natural:
                return self.central, self.context_table
            if self.parts == 1:
                return self.central, self.held[0]
            context = self.backend.empty_rows(n, ld)
            for p in sorted(self.held):
                context[p::self.parts] = self.held[p][: self.part_rows(p)]
            return self.central, context
        import torch

        assemble = root is None or comm.rank == root
        central = self.backend.empty_rows(n, ld) if assemble else None
        context = self.backend.empty_rows(n, ld) if assemble else None
        stage = self.backend.empty_rows(stripe_rows(n, 0, world), ld)
        for r in range(world):
            rows = stripe_rows(n, r, world)
            got = self._move(self.central if comm.rank == r else None, stage[:rows], r, root)
            if assemble:
                central[r::world] = got
        del stage
        # which rank holds which part now (the rotation stops anywhere): every rank learns it
        ids = sorted(self.held)
        id_t = torch.tensor(ids, dtype=torch.int64, device=self.central.device)
        owners = comm.all_gather(id_t).tolist()      # [world * per_rank], rank-major
        stage = self._spare
        for i, p in enumerate(owners):
            r = i // self.per_rank
            rows = self.part_rows(p)
            got = self._move(self.held[p][:rows] if comm.rank == r else None, stage[:rows], r, root)
            if assemble:
                context[p::self.parts] = got
        return central, context

    def _move(self, mine, stage, src: int, root: Optional[int]):
        """The tensor `mine` of rank `src` as seen by the assembling rank(s): broadcast through
        `stage` (root None) or sent to `root`."""
        comm = self.comm
        if root is None:
            if comm.rank == src:
                stage.copy_(mine)
            return comm.broadcast(stage, src)
        if src == root:
            return mine
        return comm.send_to_root(mine, root, buffer=stage, src=src)

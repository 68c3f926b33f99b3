"""Thin Python handles on the single-batch entry points of the C ABI (include/gn2v.h).

Everything stays in HBM: inputs and outputs are torch CUDA tensors used purely as device buffers
(`data_ptr()` + current stream).  uint32 node ids are carried in int32 tensors (same bits).
These are the building blocks `gn2v_train` is made of; parity tests and the
``Node2VecSequence`` batch emitter call them directly.
"""
import ctypes as C
from typing import Optional

from . import _lib
from .graph import CSRGraph


def _torch():
    import torch

    return torch


def _stream(device):
    return _torch().cuda.current_stream(device).cuda_stream


def walk_params(walk_length: int, iterations: int = 1, return_weight: float = 1.0,
                explore_weight: float = 1.0, max_neighbours: Optional[int] = 100,
                change_node_type_weight: float = 1.0, change_edge_type_weight: float = 1.0):
    for name, value in (("change_node_type_weight", change_node_type_weight),
                        ("change_edge_type_weight", change_edge_type_weight)):
        if not value > 0.0:  # 0 means "unset" only at the C boundary
            raise ValueError(f"{name} must be strictly positive, got {value}.")
    return _lib.WalkParams(walk_length, iterations, return_weight, explore_weight,
                           0 if max_neighbours is None else max_neighbours, 0,
                           change_node_type_weight, change_edge_type_weight)


def train_params(model: int, d: int, k: int, window: int, lr: float = 0.01,
                 lr_decay: float = 0.9, clip: float = 6.0, epochs: int = 1, flags: int = 1,
                 init_scale: Optional[float] = None, ld: Optional[int] = None, min_dist: int = 1):
    ld = (d + 3) // 4 * 4 if ld is None else ld
    scale = d ** -0.5 if init_scale is None else init_scale
    return _lib.TrainParams(model, d, ld, epochs, k, window, lr, lr_decay, clip, flags, scale,
                            min_dist)


def walks(graph: CSRGraph, wp, seed: int, epoch: int, first_walk: int, n_walks: int,
          device: int = 0):
    """u32 walks [n_walks, walk_length] (int32 tensor) of (seed, epoch) on the device."""
    torch = _torch()
    dg = graph.device_graph(device)
    dev = torch.device("cuda", device)
    out = torch.empty((n_walks, wp.walk_length), dtype=torch.int32, device=dev)
    _lib.check(_lib.lib().gn2v_walks(dg.handle, C.byref(wp), seed, epoch, first_walk, n_walks,
                                     out.data_ptr(), _stream(dev)))
    return out


WALK_ACCEL_EDGE_SET, WALK_ACCEL_FILTER, WALK_ACCEL_RECORDS, WALK_ACCEL_TYPED_RECORDS = 1, 2, 4, 8


def walk_accel(graph: CSRGraph, device: int = 0) -> int:
    """Bit set of the sampler's accelerators the handle holds now (gn2v_graph_walk_accel): edge
    set, its filter, edge records.  They never change a walk."""
    rc = _lib.lib().gn2v_graph_walk_accel(graph.device_graph(device).handle)
    if rc < 0:
        _lib.check(1)
    return rc


def window_batch(walks_tensor, window: int):
    """(contexts int32 [n, 2w], words int32 [n]) for every full-window walk position."""
    torch = _torch()
    n_walks, L = walks_tensor.shape
    n = n_walks * (L - 2 * window)
    dev = walks_tensor.device
    contexts = torch.empty((n, 2 * window), dtype=torch.int32, device=dev)
    words = torch.empty((n,), dtype=torch.int32, device=dev)
    _lib.check(_lib.lib().gn2v_window_batch(walks_tensor.data_ptr(), n_walks, L, window,
                                            contexts.data_ptr(), words.data_ptr(), _stream(dev)))
    return contexts, words


def walk_pairs(walks_tensor, window: int, min_dist: int = 1):
    """All (centre, context) pairs of the walks as an int32 [n_pairs, 2] tensor (uint32 bits),
    in walk / position / slot order (``gn2v_walk_pairs`` + compaction of the unused slots)."""
    torch = _torch()
    n_walks, L = walks_tensor.shape
    dev = walks_tensor.device
    slots = torch.empty((n_walks * L * 2 * window, 2), dtype=torch.int32, device=dev)
    _lib.check(_lib.lib().gn2v_walk_pairs(walks_tensor.data_ptr(), n_walks, L, window, min_dist,
                                          slots.data_ptr(), _stream(dev)))
    return slots[slots[:, 0] != -1]


def cooc_slots(walks_t, window: int, min_dist: int = 1):
    """Co-occurrence slots of the walks: (keys int64, weights int64), each [n_walks * L * 2w];
    unused slots hold (INT64_MAX, 0)."""
    torch = _torch()
    n_walks, L = walks_t.shape
    n = n_walks * L * 2 * window
    keys = torch.empty(n, dtype=torch.int64, device=walks_t.device)
    weights = torch.empty(n, dtype=torch.int64, device=walks_t.device)
    _lib.check(_lib.lib().gn2v_cooc_slots(walks_t.data_ptr(), n_walks, L, window, min_dist,
                                          keys.data_ptr(), weights.data_ptr(),
                                          _stream(walks_t.device)))
    return keys, weights


def glove_step(graph: CSRGraph, rows, cols, logx, fx, central, contextual, bias_central,
               bias_contextual, d: int, lr: float, flags: int = 0):
    """One SGD pass over the co-occurrence entries (in place on the tables and biases)."""
    dev = central.device
    dg = graph.device_graph(dev.index or 0)
    io = _lib.GloveIO(rows.data_ptr(), cols.data_ptr(), logx.data_ptr(), fx.data_ptr(),
                      central.data_ptr(), contextual.data_ptr(), bias_central.data_ptr(),
                      bias_contextual.data_ptr())
    _lib.check(_lib.lib().gn2v_glove_step(dg.handle, C.byref(io), rows.numel(), d,
                                          central.shape[1], lr, flags, _stream(dev)))


def init_table(n_rows: int, d: int, seed: int, table_id: int, scale: float, device: int = 0,
               ld: Optional[int] = None):
    torch = _torch()
    ld = (d + 3) // 4 * 4 if ld is None else ld
    dev = torch.device("cuda", device)
    t = torch.empty((n_rows, ld), dtype=torch.float32, device=dev)
    _lib.check(_lib.lib().gn2v_init_table(t.data_ptr(), n_rows, d, ld, seed, table_id, scale,
                                          _stream(dev)))
    return t


def init_table_rows(n_rows: int, d: int, seed: int, table_id: int, scale: float, first_row: int,
                    row_stride: int, device: int = 0, ld: Optional[int] = None, out=None):
    """Rows first_row, first_row + row_stride, ... of the table ``init_table`` would produce."""
    torch = _torch()
    ld = (d + 3) // 4 * 4 if ld is None else ld
    dev = torch.device("cuda", device)
    t = torch.empty((n_rows, ld), dtype=torch.float32, device=dev) if out is None else out
    assert t.is_contiguous() and t.shape == (n_rows, ld)
    _lib.check(_lib.lib().gn2v_init_table_rows(t.data_ptr(), n_rows, d, ld, seed, table_id, scale,
                                               first_row, row_stride, _stream(dev)))
    return t


# ------------------------------------------------------------- block-partitioned SkipGram
def block_plan(graph: CSRGraph, world: int, rank: int, parts: int, slices: int, walk_length: int,
               window: int, min_dist: int = 1, record: int = 32, flags: int = 0, device: int = 0,
               hot_rows: int = 0, hot_flush: int = 0):
    """A validated ``gn2v_block_plan`` (row_bits filled in by the library).  ``hot_rows``: the
    rows of every cell with the highest in-degrees that ``block_alias`` flags (at most
    ``_lib.BLOCK_HOT_MAX``; their updates are accumulated in LDS and handed over with atomics, on
    average every ``hot_flush`` updates: 0 = 16)."""
    plan = _lib.BlockPlan(world, rank, parts, slices, walk_length, window, min_dist, record, 0,
                          flags, hot_rows, hot_flush, 0, 0)
    _lib.check(_lib.lib().gn2v_block_plan_check(graph.device_graph(device).handle, C.byref(plan)))
    return plan


def block_placement(graph: CSRGraph, classes: int, seed: int, round_id: int, device: int = 0,
                    out=None):
    """(place int32 [n_nodes], inv int32 [n_nodes]) of round ``round_id``
    (``gn2v_block_placement``): the seeded permutation of the node ids inside their residue
    classes modulo ``classes`` that decides in which cell a node's contextual row is trained this
    round (place[x] = x', inv[x'] = x).  ``out``: (place, inv) tensors to fill."""
    torch = _torch()
    dg = graph.device_graph(device)
    dev = torch.device("cuda", device)
    n = graph.get_number_of_nodes()
    need = C.c_uint64()
    _lib.check(_lib.lib().gn2v_block_placement_temp_bytes(n, C.byref(need)))
    temp = torch.empty(need.value, dtype=torch.uint8, device=dev)
    place, inv = out if out is not None else (torch.empty(n, dtype=torch.int32, device=dev),
                                              torch.empty(n, dtype=torch.int32, device=dev))
    _lib.check(_lib.lib().gn2v_block_placement(dg.handle, classes, seed, round_id,
                                               place.data_ptr(), inv.data_ptr(), temp.data_ptr(),
                                               need.value, _stream(dev)))
    return place, inv


def block_place_walks(place, walks_tensor, out=None):
    """The walks with placed node ids (``gn2v_block_place_walks``): what ``block_count`` /
    ``block_extract`` read for the context side under a placement."""
    torch = _torch()
    assert walks_tensor.is_contiguous() and place.is_contiguous()
    if out is None:
        out = torch.empty_like(walks_tensor)
    _lib.check(_lib.lib().gn2v_block_place_walks(place.data_ptr(), walks_tensor.data_ptr(),
                                                 walks_tensor.numel(), out.data_ptr(),
                                                 _stream(walks_tensor.device)))
    return out


def block_alias(graph: CSRGraph, plan, device: int = 0, inv=None, out=None):
    """(alias int64 [n_nodes], cell_rows int64 [cells + 1], hub_bits int32 [(n_nodes + 31) // 32],
    hot_list int32 [cells, BLOCK_HOT_MAX], hot_slot uint8 [n_nodes]): per-cell alias tables for
    degree-proportional negatives, the hot-row flags and the hot rows' slots
    (``gn2v_block_alias``).  ``inv``: the round's placement; ``out``: (alias, cell_rows) of an
    earlier call to refill (plans without hot rows: the other three come back as None)."""
    torch = _torch()
    if out is not None or (inv is not None and not plan.hot_rows):
        dg = graph.device_graph(device)
        dev = torch.device("cuda", device)
        n = graph.get_number_of_nodes()
        need = C.c_uint64()
        _lib.check(_lib.lib().gn2v_block_alias_temp_bytes(n, C.byref(need)))
        temp = torch.empty(need.value, dtype=torch.uint8, device=dev)
        alias, cell_rows = out if out is not None else (
            torch.empty(n, dtype=torch.int64, device=dev),
            torch.empty(plan.parts * plan.slices + 1, dtype=torch.int64, device=dev))
        assert not plan.hot_rows
        _lib.check(_lib.lib().gn2v_block_alias(
            dg.handle, C.byref(plan), alias.data_ptr(), cell_rows.data_ptr(), None, None, None,
            None if inv is None else inv.data_ptr(), temp.data_ptr(), need.value, _stream(dev)))
        return alias, cell_rows, None, None, None
    dg = graph.device_graph(device)
    dev = torch.device("cuda", device)
    n = graph.get_number_of_nodes()
    need = C.c_uint64()
    _lib.check(_lib.lib().gn2v_block_alias_temp_bytes(n, C.byref(need)))
    temp = torch.empty(need.value, dtype=torch.uint8, device=dev)
    alias = torch.empty(n, dtype=torch.int64, device=dev)
    cell_rows = torch.empty(plan.parts * plan.slices + 1, dtype=torch.int64, device=dev)
    hub_bits = torch.empty((n + 31) // 32, dtype=torch.int32, device=dev)
    hot_list = torch.empty((plan.parts * plan.slices, _lib.BLOCK_HOT_MAX), dtype=torch.int32,
                           device=dev)
    hot_slot = torch.empty(n, dtype=torch.uint8, device=dev)
    _lib.check(_lib.lib().gn2v_block_alias(dg.handle, C.byref(plan), alias.data_ptr(),
                                           cell_rows.data_ptr(), hub_bits.data_ptr(),
                                           hot_list.data_ptr(), hot_slot.data_ptr(),
                                           None if inv is None else inv.data_ptr(),
                                           temp.data_ptr(), need.value, _stream(dev)))
    return alias, cell_rows, hub_bits, hot_list, hot_slot


def block_count(graph: CSRGraph, plan, walks_tensor, seed: int, epoch: int, first_walk: int,
                work=None, cell_offsets=None, part_lo: int = 0, part_n: int = 0, placed=None):
    """Pass 1 of the pair extraction: (work, cell_offsets int64 [cells + 1]); the last offset is
    the number of pairs this rank trains in the group of parts ``part_lo, part_lo + 1, ...``
    (``part_n`` of them, cyclic; 0, 0 = every part).  ``placed``: the walks with placed node ids
    (``block_place_walks``) of a round under a placement."""
    torch = _torch()
    dev = walks_tensor.device
    dg = graph.device_graph(dev.index or 0)
    if work is None:
        work = torch.empty(_lib.BLOCK_WORK_WORDS, dtype=torch.int64, device=dev)
    if cell_offsets is None:
        cell_offsets = torch.empty(plan.parts * plan.slices + 1, dtype=torch.int64, device=dev)
    assert walks_tensor.is_contiguous()
    assert placed is None or (placed.is_contiguous() and placed.shape == walks_tensor.shape)
    _lib.check(_lib.lib().gn2v_block_count(
        dg.handle, C.byref(plan), walks_tensor.data_ptr(),
        None if placed is None else placed.data_ptr(), walks_tensor.shape[0], seed, epoch,
        first_walk, part_lo, part_n, work.data_ptr(), cell_offsets.data_ptr(), _stream(dev)))
    return work, cell_offsets


def block_extract_temp_bytes(n_pairs: int) -> int:
    need = C.c_uint64()
    _lib.check(_lib.lib().gn2v_block_extract_temp_bytes(n_pairs, C.byref(need)))
    return need.value


def block_extract(graph: CSRGraph, plan, walks_tensor, seed: int, epoch: int, first_walk: int,
                  work, n_pairs: int, pairs=None, temp=None, hub_bits=None, part_lo: int = 0,
                  part_n: int = 0, placed=None):
    """Pass 2 + sort: the pair words int64 [n_pairs] (``cell << (row_bits + ctx_bits) | centre
    row << ctx_bits | hot << (ctx_bits - 1) | context row inside its cell``) grouped by cell and
    centre row."""
    torch = _torch()
    dev = walks_tensor.device
    dg = graph.device_graph(dev.index or 0)
    if pairs is None:
        pairs = torch.empty(n_pairs, dtype=torch.int64, device=dev)
    assert pairs.dtype == torch.int64 and pairs.is_contiguous()
    need = block_extract_temp_bytes(n_pairs)
    if temp is None:
        temp = torch.empty(need, dtype=torch.uint8, device=dev)
    assert pairs.numel() >= n_pairs and temp.numel() >= need
    _lib.check(_lib.lib().gn2v_block_extract(
        dg.handle, C.byref(plan), walks_tensor.data_ptr(),
        None if placed is None else placed.data_ptr(), walks_tensor.shape[0], seed, epoch,
        first_walk, part_lo, part_n, work.data_ptr(),
        None if hub_bits is None else hub_bits.data_ptr(), n_pairs, pairs.data_ptr(),
        temp.data_ptr(), temp.numel(), _stream(dev)))
    return pairs


def block_cell_offsets(graph: CSRGraph, plan, part_n: int, pairs, n_pairs: int, cell_offsets):
    """After ``block_extract``: the cell offsets of a WIDE group (more cells than the counting
    pass has LDS counters: ``gn2v_block_count`` wrote only their last entry, the number of
    pairs) read off the sorted words; does nothing for a group that was counted."""
    dev = pairs.device
    dg = graph.device_graph(dev.index or 0)
    _lib.check(_lib.lib().gn2v_block_cell_offsets(
        dg.handle, C.byref(plan), part_n, pairs.data_ptr(), n_pairs, cell_offsets.data_ptr(),
        _stream(dev)))
    return cell_offsets


def block_step(graph: CSRGraph, tp, plan, pairs, cell_offsets, alias, cell_rows, central,
               context, block_id: int, part: int, seed: int, epoch: int, lr: float,
               whole_central: bool = False, whole_context: bool = False, hot=None, inv=None,
               context_table=None):
    """Train the pairs of one context part (``gn2v_block_step``; tables updated in place).
    ``hot``: (hot_list, hot_slot) of ``block_alias`` -- with them (and ``cell_rows``) the rows the
    plan flags as hot accumulate their updates in LDS; without, they are ordinary rows.
    ``whole_central``: ``central`` is the whole table [n_nodes, ld] and the plan's rank one of its
    ``world`` centre stripes (one GPU training the stripes one after the other);
    ``whole_context``: ``context`` is the whole contextual table and the part its rows
    ``part, part + parts, ...``.  ``inv``: the round's placement (resident cells only): the rows
    of a cell are reached through it -- in ``context_table``, the whole contextual table in node
    order (``context`` may then be None), or, without, in the part's rows ``context`` (placements
    that keep the classes modulo ``parts``)."""
    dev = central.device
    dg = graph.device_graph(dev.index or 0)
    assert central.is_contiguous() and (context is None or context.is_contiguous())
    assert central.shape[1] == tp.ld and (context is None or context.shape[1] == tp.ld)
    assert context_table is None or (context_table.is_contiguous()
                                     and context_table.shape == (graph.get_number_of_nodes(), tp.ld))
    ptr = lambda t: None if t is None else t.data_ptr()  # noqa: E731
    c_ptr, c_ld, x_ptr, x_ld = ptr(central), 0, ptr(context), 0
    if whole_central:
        assert central.shape[0] == graph.get_number_of_nodes()
        c_ptr, c_ld = c_ptr + plan.rank * tp.ld * 4, plan.world * tp.ld
    if whole_context:
        assert context.shape[0] == graph.get_number_of_nodes()
        x_ptr, x_ld = x_ptr + part * tp.ld * 4, plan.parts * tp.ld
    hot_list, hot_slot = hot if hot is not None else (None, None)
    io = _lib.BlockIO(ptr(pairs), ptr(cell_offsets), ptr(alias), ptr(cell_rows), ptr(hot_list),
                      ptr(hot_slot), c_ptr, x_ptr, block_id, part, c_ld, x_ld, ptr(inv),
                      ptr(context_table))
    _lib.check(_lib.lib().gn2v_block_step(dg.handle, C.byref(tp), C.byref(plan), C.byref(io),
                                          seed, epoch, lr, _stream(dev)))


def graph_reserve_cus(graph: CSRGraph, cus_per_xcd: int, device: int = 0):
    """Leave ``cus_per_xcd`` CUs of every XCD to other work (RCCL): the training kernels of this
    graph then run on a CU-masked stream (``gn2v_graph_reserve_cus``).  Returns the CUs each XCD
    runs workgroups on under the mask, as verified by a probe launch; 0 switches it off."""
    import numpy as np

    active = np.zeros(16, dtype=np.uint32)
    _lib.check(_lib.lib().gn2v_graph_reserve_cus(graph.device_graph(device).handle, cus_per_xcd,
                                                 active.ctypes.data))
    return [int(v) for v in active if v]


def graph_xcds(graph: CSRGraph, device: int = 0) -> int:
    """XCDs (one L2 each) the device's workgroups are spread over (8 on an MI355X; 0 unknown)."""
    return _lib.lib().gn2v_graph_xcds(graph.device_graph(device).handle)


def _step(fn_name: str, graph: CSRGraph, tp, walks_tensor, seed, epoch, first_walk, lr, central,
          contextual, neg_override=None):
    dev = central.device
    dg = graph.device_graph(dev.index or 0)
    assert walks_tensor.is_contiguous() and central.is_contiguous() and contextual.is_contiguous()
    assert central.shape[1] == tp.ld and contextual.shape[1] == tp.ld
    n_walks, L = walks_tensor.shape
    fn = getattr(_lib.lib(), fn_name)
    _lib.check(fn(dg.handle, C.byref(tp), walks_tensor.data_ptr(), n_walks, L, seed, epoch,
                  first_walk, lr, central.data_ptr(), contextual.data_ptr(),
                  None if neg_override is None else neg_override.data_ptr(), _stream(dev)))


def sgns_step(graph, tp, walks_tensor, seed, epoch, first_walk, lr, central, contextual,
              neg_override=None):
    """One SkipGram negative-sampling pass over a batch of walks (tables updated in place)."""
    _step("gn2v_sgns_step", graph, tp, walks_tensor, seed, epoch, first_walk, lr, central,
          contextual, neg_override)


def cbow_step(graph, tp, walks_tensor, seed, epoch, first_walk, lr, central, contextual,
              neg_override=None):
    """One CBOW negative-sampling pass over a batch of walks (tables updated in place)."""
    _step("gn2v_cbow_step", graph, tp, walks_tensor, seed, epoch, first_walk, lr, central,
          contextual, neg_override)


def step(graph, tp, walks_tensor, seed, epoch, first_walk, lr, central, contextual,
         walk_rows=None, negative=None, neg_pool=None, neg_id_mul: int = 0, neg_id_add: int = 0,
         neg_override=None, context_delta=None):
    """General training step (``gn2v_step``): ``walk_rows`` gives the row of every walk node in
    ``central`` / ``contextual`` (compact row caches), negatives are rows of ``negative`` drawn
    from ``neg_pool``.  ``tp.model`` selects SkipGram / CBOW."""
    dev = central.device
    dg = graph.device_graph(dev.index or 0)
    n_walks, L = walks_tensor.shape
    tensors = [walks_tensor, central, contextual, walk_rows, negative, neg_pool, neg_override,
               context_delta]
    assert all(t is None or t.is_contiguous() for t in tensors)
    assert central.shape[1] == tp.ld and contextual.shape[1] == tp.ld
    ptr = lambda t: None if t is None else t.data_ptr()  # noqa: E731
    io = _lib.StepIO(ptr(walks_tensor), ptr(walk_rows), ptr(central), ptr(contextual),
                     ptr(negative), ptr(neg_pool), 0 if neg_pool is None else neg_pool.numel(),
                     neg_id_mul, neg_id_add, ptr(neg_override), ptr(context_delta))
    _lib.check(_lib.lib().gn2v_step(dg.handle, C.byref(tp), C.byref(io), n_walks, L, seed, epoch,
                                    first_walk, lr, _stream(dev)))


def touch_rows(table, ids, flags: int = 0):
    """table[ids] += 1 with the training kernels' access shape (traffic calibration)."""
    dev = table.device
    assert table.is_contiguous() and ids.is_contiguous()
    _lib.check(_lib.lib().gn2v_touch_rows(table.data_ptr(), table.shape[1], ids.data_ptr(),
                                          ids.numel(), flags, _stream(dev)))


def stats_reset(graph: CSRGraph, device: int = 0):
    dg = graph.device_graph(device)
    _lib.check(_lib.lib().gn2v_stats_reset(dg.handle, _stream(_torch().device("cuda", device))))


def stats_read(graph: CSRGraph, device: int = 0) -> dict:
    dg = graph.device_graph(device)
    st = _lib.Stats()
    _lib.check(_lib.lib().gn2v_stats_read(dg.handle, C.byref(st),
                                          _stream(_torch().device("cuda", device))))
    return st.as_dict()

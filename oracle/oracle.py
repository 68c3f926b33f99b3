"""ctypes front-end of the CPU oracle (``oracle/gn2v_oracle.c``).

TEST INFRASTRUCTURE ONLY -- imported by ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py``; never by ``embiggen_amd``.  PARITY UNPINNED: see the header
of ``gn2v_oracle.c`` (the reference's arithmetic lives in the un-vendored ``ensmallen`` wheel,
``embiggen/embedders/ensmallen_embedders/node2vec.py:99``).

Everything takes / returns plain numpy arrays.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
SENTINEL = 0xFFFFFFFF

FLAG_SCALE_FREE = 1
FLAG_DOWNSAMPLE = 2
FLAG_NORM_LR = 4
# oracle-only variants of two semantic readings (gn2v_oracle.c O_FLAG_SKIP_CLIPPED / _SHARED_NEGATIVES)
FLAG_SKIP_CLIPPED = 256
FLAG_SHARED_NEGATIVES = 512


class WalkParams(C.Structure):
    _fields_ = [
        ("walk_length", C.c_uint32),
        ("iterations", C.c_uint32),
        ("return_weight", C.c_float),
        ("explore_weight", C.c_float),
        ("max_neighbours", C.c_uint32),
        ("flags", C.c_uint32),
        ("change_node_type_weight", C.c_float),  # 0 = unset = 1.0
        ("change_edge_type_weight", C.c_float),
    ]


class TrainParams(C.Structure):
    _fields_ = [
        ("model", C.c_uint32),
        ("d", C.c_uint32),
        ("ld", C.c_uint32),
        ("epochs", C.c_uint32),
        ("k", C.c_uint32),
        ("window", C.c_uint32),
        ("lr", C.c_float),
        ("lr_decay", C.c_float),
        ("clip", C.c_float),
        ("flags", C.c_uint32),
        ("init_scale", C.c_float),
        ("min_dist", C.c_uint32),
    ]


class Graph(C.Structure):
    _fields_ = [
        ("n_nodes", C.c_uint64),
        ("n_edges", C.c_uint64),
        ("row_ptr", C.c_void_p),
        ("col_idx", C.c_void_p),
        ("cumw", C.c_void_p),
        ("node_types", C.c_void_p),
        ("edge_types", C.c_void_p),
    ]


def build(force: bool = False, asan: bool = False, fast: bool = False) -> str:
    """Compile the oracle with gcc (idempotent) and return the path of the shared object.
    ``fast``: the tuned build of the same source (``-ffast-math -DO_FAST``: vectorised dot
    products, software prefetch) that only ``bench.py``'s ``cpu_baseline`` times; with neither
    flag both the strict and the tuned library are brought up to date."""
    srcs = [os.path.join(_HERE, "gn2v_oracle.c"), os.path.join(_HERE, "gn2v_cpu.c"),
            os.path.join(_HERE, "gn2v_cpu.h"),
            os.path.join(os.path.dirname(_HERE), "include", "gn2v.h")]
    newest = max(os.path.getmtime(f) for f in srcs)
    names = (["libgn2v_oracle_asan.so"] if asan else ["libgn2v_oracle_fast.so"] if fast
             else ["libgn2v_oracle.so", "libgn2v_oracle_fast.so"])
    for target in names:
        path = os.path.join(_HERE, target)
        if force or not os.path.exists(path) or os.path.getmtime(path) < newest:
            subprocess.run(["make", "-C", _HERE, target], check=True, capture_output=True)
    return os.path.join(_HERE, names[0])


_lib = None
_fast_lib = None


def _declare(L):
    L.o_mix64.restype = C.c_uint64
    L.o_mix64.argtypes = [C.c_uint64]
    L.o_draw.restype = C.c_uint64
    L.o_draw.argtypes = [C.c_uint64, C.c_uint64]
    L.o_walk_key.restype = C.c_uint64
    L.o_walk_key.argtypes = [C.c_uint64, C.c_uint64, C.c_uint64]
    L.o_fit.restype = C.c_uint64
    L.o_window_batch.restype = C.c_uint64
    L.o_walk_pairs.restype = C.c_uint64
    L.gn2v_cpu_last_error.restype = C.c_char_p
    return L


def lib():
    global _lib
    if _lib is None:
        _lib = _declare(C.CDLL(build()))
    return _lib


def fast_lib():
    """The tuned build (timed CPU baseline only; never the checker)."""
    global _fast_lib
    if _fast_lib is None:
        _fast_lib = _declare(C.CDLL(build(fast=True)))
    return _fast_lib


def _ptr(a):
    return None if a is None else C.c_void_p(a.ctypes.data)


class OracleGraph:
    """Holds CSR arrays (kept alive) and the C struct that points at them."""

    def __init__(self, row_ptr, col_idx, cumw=None, node_types=None, edge_types=None):
        self.row_ptr = np.ascontiguousarray(row_ptr, dtype=np.uint64)
        self.col_idx = np.ascontiguousarray(col_idx, dtype=np.uint32)
        self.cumw = None if cumw is None else np.ascontiguousarray(cumw, dtype=np.float32)
        self.node_types = (None if node_types is None
                           else np.ascontiguousarray(node_types, dtype=np.uint32))
        self.edge_types = (None if edge_types is None
                           else np.ascontiguousarray(edge_types, dtype=np.uint32))
        self.n_nodes = len(self.row_ptr) - 1
        self.n_edges = len(self.col_idx)
        self.c = Graph(
            self.n_nodes,
            self.n_edges,
            self.row_ptr.ctypes.data,
            self.col_idx.ctypes.data,
            None if self.cumw is None else self.cumw.ctypes.data,
            None if self.node_types is None else self.node_types.ctypes.data,
            None if self.edge_types is None else self.edge_types.ctypes.data,
        )


def mix64(z: int) -> int:
    return lib().o_mix64(C.c_uint64(z & (2**64 - 1)))


def ba_edges(n_nodes: int, m: int, seed: int):
    n_e = (n_nodes - 1) * m
    src = np.empty(n_e, dtype=np.uint32)
    dst = np.empty(n_e, dtype=np.uint32)
    lib().o_ba_edges(C.c_uint64(n_nodes), C.c_uint32(m), C.c_uint64(seed), _ptr(src), _ptr(dst))
    return src, dst


def walks(g: OracleGraph, wp: WalkParams, seed: int, epoch: int, first_walk: int, n_walks: int,
          sources=None):
    if sources is None:
        n_sources = g.n_nodes
    else:
        sources = np.ascontiguousarray(sources, dtype=np.uint32)
        n_sources = len(sources)
    out = np.empty((n_walks, wp.walk_length), dtype=np.uint32)
    lib().o_walks(C.byref(g.c), C.byref(wp), _ptr(sources), C.c_uint64(n_sources),
                  C.c_uint64(seed), C.c_uint64(epoch), C.c_uint64(first_walk),
                  C.c_uint64(n_walks), _ptr(out))
    return out


def init_table(n_rows: int, d: int, ld: int, seed: int, table_id: int, scale: float):
    t = np.empty((n_rows, ld), dtype=np.float32)
    lib().o_init_table(_ptr(t), C.c_uint64(n_rows), C.c_uint32(d), C.c_uint32(ld),
                       C.c_uint64(seed), C.c_uint32(table_id), C.c_float(scale))
    return t


def train_walks(g: OracleGraph, tp: TrainParams, walks_arr, seed: int, epoch: int,
                first_walk: int, lr: float, central, contextual, neg_override=None,
                threads: int = 1, fast: bool = False):
    """In-place update of ``central`` / ``contextual`` ([N, ld] float32, C-contiguous).
    ``fast``: through the tuned build (the timed CPU baseline)."""
    walks_arr = np.ascontiguousarray(walks_arr, dtype=np.uint32)
    assert central.flags.c_contiguous and contextual.flags.c_contiguous
    assert central.dtype == np.float32 and contextual.dtype == np.float32
    if neg_override is not None:
        neg_override = np.ascontiguousarray(neg_override, dtype=np.uint32)
    n_walks, L = walks_arr.shape
    (fast_lib() if fast else lib()).o_train_walks(
                        C.byref(g.c), C.byref(tp), _ptr(walks_arr), C.c_uint64(n_walks),
                        C.c_uint32(L), C.c_uint64(seed), C.c_uint64(epoch),
                        C.c_uint64(first_walk), C.c_float(lr), _ptr(central), _ptr(contextual),
                        _ptr(neg_override), C.c_int(threads))


class StepIO(C.Structure):
    _fields_ = [
        ("walks", C.c_void_p),
        ("walk_rows", C.c_void_p),
        ("central", C.c_void_p),
        ("contextual", C.c_void_p),
        ("negative", C.c_void_p),
        ("neg_pool", C.c_void_p),
        ("neg_pool_size", C.c_uint64),
        ("neg_id_mul", C.c_uint32),
        ("neg_id_add", C.c_uint32),
        ("neg_override", C.c_void_p),
    ]


def train_walks_ex(g: OracleGraph, tp: TrainParams, walks_arr, seed: int, epoch: int,
                   first_walk: int, lr: float, central, contextual, walk_rows=None,
                   negative=None, neg_pool=None, neg_id_mul: int = 0, neg_id_add: int = 0,
                   neg_override=None, threads: int = 1):
    """General step (mirrors gn2v_step): walk nodes addressed through ``walk_rows``, negatives
    drawn from ``neg_pool`` as rows of ``negative``.  All arrays are updated in place."""
    walks_arr = np.ascontiguousarray(walks_arr, dtype=np.uint32)
    keep = [walks_arr]
    io = StepIO()
    io.walks = walks_arr.ctypes.data
    for name, arr, dtype in (("walk_rows", walk_rows, np.uint32), ("neg_pool", neg_pool, np.uint32),
                             ("neg_override", neg_override, np.uint32)):
        if arr is not None:
            arr = np.ascontiguousarray(arr, dtype=dtype)
            keep.append(arr)
            setattr(io, name, arr.ctypes.data)
    for name, arr in (("central", central), ("contextual", contextual), ("negative", negative)):
        if arr is not None:
            assert arr.flags.c_contiguous and arr.dtype == np.float32
            setattr(io, name, arr.ctypes.data)
    io.neg_pool_size = 0 if neg_pool is None else int(np.asarray(neg_pool).size)
    io.neg_id_mul, io.neg_id_add = neg_id_mul, neg_id_add
    n_walks, L = walks_arr.shape
    lib().o_train_walks_ex(C.byref(g.c), C.byref(tp), C.byref(io), C.c_uint64(n_walks),
                           C.c_uint32(L), C.c_uint64(seed), C.c_uint64(epoch),
                           C.c_uint64(first_walk), C.c_float(lr), C.c_int(threads))


def fit(g: OracleGraph, wp: WalkParams, tp: TrainParams, seed: int, sources=None,
        threads: int = 1):
    """Full fit_transform restatement -> (central, contextual, n_pairs)."""
    if sources is None:
        n_sources = g.n_nodes
    else:
        sources = np.ascontiguousarray(sources, dtype=np.uint32)
        n_sources = len(sources)
    central = np.empty((g.n_nodes, tp.ld), dtype=np.float32)
    contextual = np.empty((g.n_nodes, tp.ld), dtype=np.float32)
    pairs = lib().o_fit(C.byref(g.c), C.byref(wp), C.byref(tp), _ptr(sources),
                        C.c_uint64(n_sources), C.c_uint64(seed), _ptr(central),
                        _ptr(contextual), C.c_int(threads))
    return central, contextual, int(pairs)


def window_batch(walks_arr, window: int):
    walks_arr = np.ascontiguousarray(walks_arr, dtype=np.uint32)
    n_walks, L = walks_arr.shape
    n = n_walks * max(L - 2 * window, 0)
    contexts = np.empty((n, 2 * window), dtype=np.int32)
    words = np.empty(n, dtype=np.int32)
    got = lib().o_window_batch(_ptr(walks_arr), C.c_uint64(n_walks), C.c_uint32(L),
                               C.c_uint32(window), _ptr(contexts), _ptr(words))
    assert got == n
    return contexts, words


def walk_pairs(walks_arr, window: int, min_dist: int = 1):
    """(centre, context) pairs uint32 [n, 2] in walk / position / slot order."""
    walks_arr = np.ascontiguousarray(walks_arr, dtype=np.uint32)
    n_walks, L = walks_arr.shape
    out = np.empty((n_walks * L * 2 * window, 2), dtype=np.uint32)
    n = lib().o_walk_pairs(_ptr(walks_arr), C.c_uint64(n_walks), C.c_uint32(L),
                           C.c_uint32(window), C.c_uint32(min_dist), _ptr(out))
    return out[:n].copy()


# ------------------------------------------------------------------------------------------ GloVe
COOC_UNUSED = 0x7FFFFFFFFFFFFFFF


def cooc_slots(walks_arr, window: int, min_dist: int = 1):
    """(keys u64, weights u64) per slot [n_walks * L * 2w]; unused slots hold COOC_UNUSED / 0."""
    walks_arr = np.ascontiguousarray(walks_arr, dtype=np.uint32)
    n_walks, L = walks_arr.shape
    n = n_walks * L * 2 * window
    keys, weights = np.empty(n, dtype=np.uint64), np.empty(n, dtype=np.uint64)
    lib().o_cooc_slots(_ptr(walks_arr), C.c_uint64(n_walks), C.c_uint32(L), C.c_uint32(window),
                       C.c_uint32(min_dist), _ptr(keys), _ptr(weights))
    return keys, weights


def cooc_reduce(keys, weights):
    """Distinct keys ascending and their summed fixed-point counts."""
    keys = np.array(keys, dtype=np.uint64)
    weights = np.array(weights, dtype=np.uint64)
    lib().o_cooc_reduce.restype = C.c_uint64
    n = lib().o_cooc_reduce(_ptr(keys), _ptr(weights), C.c_uint64(len(keys)))
    return keys[:n].copy(), weights[:n].copy()


GLOVE_RECORD = 16


def glove_entries(keys, counts, seed: int, alpha: float):
    """Slots of the training records (rows, cols, log X, f(X)): GLOVE_RECORD consecutive slots
    share their row; padding slots have col == SENTINEL."""
    keys = np.ascontiguousarray(keys, dtype=np.uint64)
    counts = np.ascontiguousarray(counts, dtype=np.uint64)
    lib().o_glove_record_count.restype = C.c_uint64
    n = int(lib().o_glove_record_count(_ptr(keys), C.c_uint64(len(keys)))) * GLOVE_RECORD
    rows, cols = np.empty(n, dtype=np.uint32), np.empty(n, dtype=np.uint32)
    logx, fx = np.empty(n, dtype=np.float32), np.empty(n, dtype=np.float32)
    lib().o_glove_entries(_ptr(keys), _ptr(counts), C.c_uint64(len(keys)), C.c_uint64(seed),
                          C.c_float(alpha), _ptr(rows), _ptr(cols), _ptr(logx), _ptr(fx))
    return rows, cols, logx, fx


def glove_step(rows, cols, logx, fx, central, contextual, bias_c, bias_x, d: int, lr: float):
    """In-place sequential SGD over the entries."""
    for a, t in ((rows, np.uint32), (cols, np.uint32), (logx, np.float32), (fx, np.float32),
                 (central, np.float32), (contextual, np.float32), (bias_c, np.float32),
                 (bias_x, np.float32)):
        assert a.dtype == t and a.flags.c_contiguous
    lib().o_glove_step(_ptr(rows), _ptr(cols), _ptr(logx), _ptr(fx), C.c_uint64(len(rows)),
                       _ptr(central), _ptr(contextual), _ptr(bias_c), _ptr(bias_x),
                       C.c_uint32(d), C.c_uint32(central.shape[1]), C.c_float(lr))


def glove_step_rounds(rows, cols, logx, fx, central, contextual, bias_c, bias_x, d: int, lr: float):
    """In-place SGD in the engine's record schedule (four entries of a record per round)."""
    lib().o_glove_step_rounds(_ptr(rows), _ptr(cols), _ptr(logx), _ptr(fx), C.c_uint64(len(rows)),
                              _ptr(central), _ptr(contextual), _ptr(bias_c), _ptr(bias_x),
                              C.c_uint32(d), C.c_uint32(central.shape[1]), C.c_float(lr))


def glove_loss(rows, cols, logx, fx, central, contextual, bias_c, bias_x, d: int) -> float:
    lib().o_glove_loss.restype = C.c_double
    return lib().o_glove_loss(_ptr(rows), _ptr(cols), _ptr(logx), _ptr(fx), C.c_uint64(len(rows)),
                              _ptr(central), _ptr(contextual), _ptr(bias_c), _ptr(bias_x),
                              C.c_uint32(d), C.c_uint32(central.shape[1]))


# ------------------------------------------------------------- block-partitioned SkipGram
class BlockPlan(C.Structure):
    """Mirror of gn2v_block_plan (include/gn2v.h)."""

    _fields_ = [(name, C.c_uint32) for name in (
        "world", "rank", "parts", "slices", "walk_length", "window", "min_dist", "record",
        "row_bits", "flags", "hot_rows", "hot_flush", "key_bits", "ctx_bits")]


def block_plan(n_nodes: int, world: int, rank: int, parts: int, slices: int, walk_length: int,
               window: int, min_dist: int = 1, record: int = 16, flags: int = 0,
               hot_rows: int = 0) -> BlockPlan:
    L = lib()
    for fn in (L.o_block_row_bits, L.o_block_ctx_bits, L.o_block_cell_bits):
        fn.restype = C.c_uint32
    row_bits = L.o_block_row_bits(C.c_uint64(n_nodes), C.c_uint32(world))
    ctx_bits = L.o_block_ctx_bits(C.c_uint64(n_nodes), C.c_uint32(parts), C.c_uint32(slices))
    cell_bits = L.o_block_cell_bits(C.c_uint32(parts), C.c_uint32(slices))
    return BlockPlan(world, rank, parts, slices, walk_length, window, min_dist, record, row_bits,
                     flags, hot_rows, 0, cell_bits + row_bits + ctx_bits, ctx_bits)


def block_placement(n_nodes: int, classes: int, seed: int, round_id: int):
    """(place u32[n_nodes], inv u32[n_nodes]): the round's seeded permutation of the node ids
    inside their residue classes modulo ``classes`` (place[x] = x', inv[x'] = x)."""
    place = np.empty(n_nodes, dtype=np.uint32)
    inv = np.empty(n_nodes, dtype=np.uint32)
    lib().o_block_placement(C.c_uint64(n_nodes), C.c_uint32(classes), C.c_uint64(seed),
                            C.c_uint64(round_id), _ptr(place), _ptr(inv))
    return place, inv


RESIDENT_CENTRE_SORT_BITS = 8  # GN2V_RESIDENT_CENTRE_SORT_BITS (include/gn2v.h)


def block_sort_shift(plan: BlockPlan) -> int:
    """The pair words of a group are sorted stably on the bits from here up: (cell, centre row)
    for XCD plans, (cell, the centre row's highest RESIDENT_CENTRE_SORT_BITS bits) for resident
    plans (more than 16 slices) -- include/gn2v.h GN2V_RESIDENT_CENTRE_SORT_BITS."""
    if plan.slices > 16 and plan.row_bits > RESIDENT_CENTRE_SORT_BITS:
        return plan.ctx_bits + plan.row_bits - RESIDENT_CENTRE_SORT_BITS
    return plan.ctx_bits


def block_extract(g: OracleGraph, plan: BlockPlan, walks_arr, seed: int, epoch: int,
                  first_walk: int, sort: bool = True, hub_bits=None, part_lo: int = 0,
                  part_n: int = 0, place=None):
    """(words u64, cell_offsets): the pairs of the walks whose centre `plan.rank` owns (contexts
    in the parts part_lo, part_lo + 1, ... cyclic; 0, 0 = all) as pair words
    ``cell << (row_bits + ctx_bits) | centre row << ctx_bits | hot << (ctx_bits - 1) | context
    row inside its cell``, sorted stably on the bits from block_sort_shift(plan) up
    (``sort=False``: extraction order)."""
    walks_arr = np.ascontiguousarray(walks_arr, dtype=np.uint32)
    n_walks = walks_arr.shape[0]
    L = lib()
    L.o_block_extract.restype = C.c_uint64
    args = (C.byref(g.c), C.byref(plan), _ptr(walks_arr), C.c_uint64(n_walks), C.c_uint64(seed),
            C.c_uint64(epoch), C.c_uint64(first_walk), C.c_uint32(part_lo), C.c_uint32(part_n))
    if place is not None:
        assert place.dtype == np.uint32 and place.flags.c_contiguous
    n = int(L.o_block_extract(*args, None, None, _ptr(place)))
    words = np.empty(n, dtype=np.uint64)
    if n:
        L.o_block_extract(*args, _ptr(hub_bits), _ptr(words), _ptr(place))
        if sort:
            L.o_block_sort(_ptr(words), C.c_uint64(n), C.c_uint32(block_sort_shift(plan)))
    cells = plan.parts * plan.slices
    offsets = np.zeros(cells + 1, dtype=np.uint64)
    if sort:
        L.o_block_cell_offsets(_ptr(words), C.c_uint64(n),
                               C.c_uint32(plan.row_bits + plan.ctx_bits), C.c_uint32(cells),
                               _ptr(offsets))
    return words, offsets


def block_unpack(words, plan: BlockPlan):
    """Pair words -> (cell, centre row, context row inside its PART, hot flag) arrays."""
    words = np.asarray(words, dtype=np.uint64)
    low = words & np.uint64((1 << plan.ctx_bits) - 1)
    hot = (low >> np.uint64(plan.ctx_bits - 1)).astype(np.uint32)
    local = (low & np.uint64((1 << (plan.ctx_bits - 1)) - 1)).astype(np.uint32)
    key = words >> np.uint64(plan.ctx_bits)
    cell = (key >> np.uint64(plan.row_bits)).astype(np.uint32)
    crow = (key & np.uint64((1 << plan.row_bits) - 1)).astype(np.uint32)
    row = (cell % np.uint32(plan.slices)) + np.uint32(plan.slices) * local
    return cell, crow, row, hot


def block_pack(cell, centre_row, part_row, plan: BlockPlan, hot=None):
    """The inverse: pair words from cells, centre rows and context rows inside their part (the
    row's slice must be the cell's: part_row % slices == cell % slices)."""
    cell = np.asarray(cell, dtype=np.uint64)
    part_row = np.asarray(part_row, dtype=np.uint64)
    assert ((part_row % np.uint64(plan.slices)) == (cell % np.uint64(plan.slices))).all()
    local = part_row // np.uint64(plan.slices)
    hot = np.zeros(len(part_row), dtype=np.uint64) if hot is None else np.asarray(hot, np.uint64)
    key = (cell << np.uint64(plan.row_bits)) | np.asarray(centre_row, dtype=np.uint64)
    return (key << np.uint64(plan.ctx_bits)) | (hot << np.uint64(plan.ctx_bits - 1)) | local


HOT_MAX = 192  # O_HOT_MAX / GN2V_BLOCK_HOT_MAX


def block_alias(g: OracleGraph, parts: int, slices: int, hot_rows: int = 0, inv=None):
    """(alias tables u64[n_nodes], cell_rows u64[cells + 1], hub_bits u32[(n_nodes + 31) // 32],
    hot_list u32[cells, HOT_MAX], hot_slot u8[n_nodes]):
    entry = hot(row) | threshold (31 bits) | alias row << 32 | hot(alias row) << 63; the hot rows
    of a cell are its ``hot_rows`` rows of highest in-degree (hot_list: slot -> row inside the
    cell, hot_slot: entry -> slot or 0xFF)."""
    table = np.zeros(g.n_nodes, dtype=np.uint64)
    cell_rows = np.empty(parts * slices + 1, dtype=np.uint64)
    hub_bits = np.zeros((g.n_nodes + 31) // 32, dtype=np.uint32)
    hot_list = np.empty((parts * slices, HOT_MAX), dtype=np.uint32)
    hot_slot = np.empty(g.n_nodes, dtype=np.uint8)
    lib().o_block_alias(C.byref(g.c), C.c_uint32(parts), C.c_uint32(slices), C.c_uint32(hot_rows),
                        _ptr(table), _ptr(cell_rows), _ptr(hub_bits), _ptr(hot_list),
                        _ptr(hot_slot), _ptr(inv))
    return table, cell_rows, hub_bits, hot_list, hot_slot


def block_step(g: OracleGraph, tp: TrainParams, plan: BlockPlan, words, cell_offsets, alias,
               cell_rows, central, context, block_id: int, part: int, seed: int, epoch: int,
               lr: float, inv=None, natural: bool = False) -> int:
    """Sequential training of one part (in place on ``central`` / ``context``).  ``inv``: the
    round's placement (``block_placement``); ``natural``: ``context`` is the whole contextual
    table in node order instead of the rows of the part."""
    if inv is not None:
        assert inv.dtype == np.uint32 and inv.flags.c_contiguous
    for a, t in ((words, np.uint64), (cell_offsets, np.uint64), (central, np.float32),
                 (context, np.float32)):
        assert a.dtype == t and a.flags.c_contiguous
    lib().o_block_step.restype = C.c_uint64
    return int(lib().o_block_step(
        C.byref(g.c), C.byref(tp), C.byref(plan), _ptr(words), _ptr(cell_offsets),
        _ptr(alias), _ptr(cell_rows), _ptr(central), _ptr(context), C.c_uint64(block_id),
        C.c_uint32(part), C.c_uint64(seed), C.c_uint64(epoch), C.c_float(lr), _ptr(inv),
        C.c_uint32(1 if natural else 0)))


def block_negatives(g: OracleGraph, tp: TrainParams, plan: BlockPlan, words, cell_offsets, alias,
                    cell_rows, block_id: int, part: int, seed: int, epoch: int, inv=None):
    """Test hook: the negatives ``block_step`` trains for the pairs of ``part`` -- node ids
    [pairs of the part, k], 0xFFFFFFFF where a sample was given up -- in the sorted order of
    ``words``."""
    lo = int(cell_offsets[part * plan.slices])
    hi = int(cell_offsets[(part + 1) * plan.slices])
    out = np.full((hi - lo, max(1, tp.k)), 0xFFFFFFFF, dtype=np.uint32)
    lib().o_block_negatives.restype = C.c_uint64
    n = int(lib().o_block_negatives(
        C.byref(g.c), C.byref(tp), C.byref(plan), _ptr(words), _ptr(cell_offsets), _ptr(alias),
        _ptr(cell_rows), C.c_uint64(block_id), C.c_uint32(part), C.c_uint64(seed),
        C.c_uint64(epoch), _ptr(inv), _ptr(out)))
    assert n == hi - lo
    return out[:, :tp.k]


def init_table_rows(n_rows: int, d: int, ld: int, seed: int, table_id: int, scale: float,
                    first_row: int, row_stride: int):
    t = np.empty((n_rows, ld), dtype=np.float32)
    lib().o_init_table_rows(_ptr(t), C.c_uint64(n_rows), C.c_uint32(d), C.c_uint32(ld),
                            C.c_uint64(seed), C.c_uint32(table_id), C.c_float(scale),
                            C.c_uint64(first_row), C.c_uint64(row_stride))
    return t


def block_record_stride(n_records: int) -> int:
    lib().o_block_record_stride.restype = C.c_uint64
    return int(lib().o_block_record_stride(C.c_uint64(n_records)))


# ---------------------------------------------------------------------------------------------
# The CPU twins of include/gn2v.h's compute entry points (oracle/gn2v_cpu.h): same argument
# lists as the device library's, host arrays in place of device pointers.


class CpuTwinError(RuntimeError):
    pass


def cpu_check(rc: int):
    if rc != 0:
        raise CpuTwinError(lib().gn2v_cpu_last_error().decode())


class CpuGraph:
    """gn2v_cpu_graph over host CSR arrays (borrowed: kept alive here).  ``threads`` <= 1: the
    sequential checker; > 1: Hogwild over walks."""

    def __init__(self, row_ptr, col_idx, cumw=None, sources=None, threads: int = 1):
        self.row_ptr = np.ascontiguousarray(row_ptr, dtype=np.uint64)
        self.col_idx = np.ascontiguousarray(col_idx, dtype=np.uint32)
        self.cumw = None if cumw is None else np.ascontiguousarray(cumw, dtype=np.float32)
        self.sources = None if sources is None else np.ascontiguousarray(sources, dtype=np.uint32)
        self.n_nodes = len(self.row_ptr) - 1
        self.handle = C.c_void_p()
        self._types = (None, None)
        cpu_check(lib().gn2v_cpu_graph_create(
            _ptr(self.row_ptr), _ptr(self.col_idx), _ptr(self.cumw), _ptr(self.sources),
            C.c_uint64(self.n_nodes), C.c_uint64(len(self.col_idx)),
            C.c_uint64(0 if self.sources is None else len(self.sources)), C.c_uint32(0),
            C.c_int(threads), C.byref(self.handle)))

    def set_types(self, node_types=None, edge_types=None):
        nt = None if node_types is None else np.ascontiguousarray(node_types, dtype=np.uint32)
        et = None if edge_types is None else np.ascontiguousarray(edge_types, dtype=np.uint32)
        self._types = (nt, et)
        cpu_check(lib().gn2v_cpu_graph_set_types(self.handle, _ptr(nt), _ptr(et)))

    def __del__(self):
        if getattr(self, "handle", None):
            lib().gn2v_cpu_graph_destroy(self.handle)
            self.handle = None


def cpu_walks(g: CpuGraph, wp, seed: int, epoch: int, first_walk: int, n_walks: int):
    """gn2v_cpu_walks: the arguments of gn2v_walks (``wp`` any struct of gn2v_walk_params' layout)."""
    out = np.empty((n_walks, wp.walk_length), dtype=np.uint32)
    cpu_check(lib().gn2v_cpu_walks(g.handle, C.byref(wp), C.c_uint64(seed), C.c_uint64(epoch),
                                   C.c_uint64(first_walk), C.c_uint64(n_walks), _ptr(out), None))
    return out


def cpu_walks_strided(g: CpuGraph, wp, seed: int, epoch: int, first_walk: int, n_walks: int,
                      group: int, stride: int):
    """gn2v_cpu_walks_strided: walk b has id first_walk + (b // group) * stride + b % group."""
    out = np.empty((n_walks, wp.walk_length), dtype=np.uint32)
    cpu_check(lib().gn2v_cpu_walks_strided(g.handle, C.byref(wp), C.c_uint64(seed),
                                           C.c_uint64(epoch), C.c_uint64(first_walk),
                                           C.c_uint64(n_walks), C.c_uint32(group),
                                           C.c_uint64(stride), _ptr(out), None))
    return out


def cpu_window_batch(walks_arr, window: int):
    wk = np.ascontiguousarray(walks_arr, dtype=np.uint32)
    n = wk.shape[0] * (wk.shape[1] - 2 * window)
    contexts = np.empty((max(n, 0), 2 * window), dtype=np.int32)
    words = np.empty((max(n, 0),), dtype=np.int32)
    cpu_check(lib().gn2v_cpu_window_batch(_ptr(wk), C.c_uint64(wk.shape[0]), C.c_uint32(wk.shape[1]),
                                          C.c_uint32(window), _ptr(contexts), _ptr(words), None))
    return contexts, words


def cpu_init_table(n_rows: int, d: int, ld: int, seed: int, table_id: int, scale: float):
    t = np.empty((n_rows, ld), dtype=np.float32)
    cpu_check(lib().gn2v_cpu_init_table(_ptr(t), C.c_uint64(n_rows), C.c_uint32(d), C.c_uint32(ld),
                                        C.c_uint64(seed), C.c_uint32(table_id), C.c_float(scale),
                                        None))
    return t


def cpu_step(g: CpuGraph, tp, model: int, walks_arr, seed: int, epoch: int, first_walk: int,
             lr: float, central, contextual, neg_override=None):
    """gn2v_cpu_sgns_step (model 0) / gn2v_cpu_cbow_step (model 1): tables updated in place."""
    wk = np.ascontiguousarray(walks_arr, dtype=np.uint32)
    ov = None if neg_override is None else np.ascontiguousarray(neg_override, dtype=np.uint32)
    fn = lib().gn2v_cpu_cbow_step if model else lib().gn2v_cpu_sgns_step
    cpu_check(fn(g.handle, C.byref(tp), _ptr(wk), C.c_uint64(wk.shape[0]), C.c_uint32(wk.shape[1]),
                 C.c_uint64(seed), C.c_uint64(epoch), C.c_uint64(first_walk), C.c_float(lr),
                 _ptr(central), _ptr(contextual), _ptr(ov), None))


def cpu_train(g: CpuGraph, wp, tp, seed: int, max_walks_per_epoch: int = 0, stats=None):
    """gn2v_cpu_train: (central, contextual) float32 [n_nodes, ld]; ``stats`` (optional) any
    struct of gn2v_stats' layout, filled with pairs / walk_steps / centres."""
    central = np.empty((g.n_nodes, tp.ld), dtype=np.float32)
    contextual = np.empty((g.n_nodes, tp.ld), dtype=np.float32)
    cpu_check(lib().gn2v_cpu_train(g.handle, C.byref(wp), C.byref(tp), C.c_uint64(seed),
                                   C.c_uint64(max_walks_per_epoch), _ptr(central), _ptr(contextual),
                                   None if stats is None else C.byref(stats), None))
    return central, contextual

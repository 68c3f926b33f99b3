"""ctypes binding of ``libgn2v.so`` (C ABI declared in ``include/gn2v.h``).

This is the seam that replaces the PyO3 boundary of the reference,
``self._model.fit_transform(graph)`` (embiggen/embedders/ensmallen_embedders/node2vec.py:99).
There is no CPU fallback: if the shared object is missing or no AMD GPU is visible the engine
raises, it never silently computes elsewhere.
"""
import ctypes as C
import os
import subprocess

_CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")
LIB_PATH = os.path.join(_CSRC, "libgn2v.so")
_UNITS = ["gn2v_api.hip", "gn2v_block_api.hip", "gn2v_rccl.hip"]  # translation units of libgn2v.so
_HEADER = os.path.join(os.path.dirname(_CSRC), "..", "include", "gn2v.h")
_HEADER_EXPERIMENTAL = os.path.join(os.path.dirname(_CSRC), "..", "include", "gn2v_experimental.h")
_HEADER_INTERNAL = os.path.join(os.path.dirname(_CSRC), "..", "include", "gn2v_internal.h")
_HEADER_RCCL = os.path.join(os.path.dirname(_CSRC), "..", "include", "gn2v_rccl.h")

SENTINEL = 0xFFFFFFFF
GRAPH_DEVICE_PTRS = 1
GRAPH_SYMMETRIC = 2
TRAIN_SCALE_FREE = 1
TRAIN_DOWNSAMPLE = 2
TRAIN_NORM_LR = 4
TRAIN_DETERMINISTIC = 8
TRAIN_ATOMIC = 16
TRAIN_WRITE_BACK = 32
TRAIN_WRITE_THROUGH = 64
TRAIN_NO_CTX_CACHE = 128
TRAIN_CTX_CACHE_ALL = 256
BLOCK_PATH_MIN_NODES = 2560  # GN2V_BLOCK_PATH_MIN_NODES
TRAIN_WALK_ORDERED = 1024
TRAIN_BLOCK_PATH = 2048
TRAIN_CENTRAL_STORE = 4096
TRAIN_CTX_CACHE_NONE = 512
MODEL_SKIPGRAM = 0
MODEL_CBOW = 1

# every symbol include/gn2v.h declares -- the drop-in boundary (checked by tests/test_cabi.py)
BOUNDARY_EXPORTS = [
    "gn2v_version", "gn2v_last_error", "gn2v_device_count", "gn2v_graph_create",
    "gn2v_graph_destroy", "gn2v_graph_set_types", "gn2v_ba_edges", "gn2v_walks",
    "gn2v_walks_strided",
    "gn2v_window_batch", "gn2v_walk_pairs", "gn2v_init_table",
    "gn2v_sgns_step", "gn2v_cbow_step", "gn2v_train", "gn2v_train_blocks", "gn2v_train_world",
    "gn2v_edge_embedding", "gn2v_cooc_slots", "gn2v_glove_step",
    "gn2v_graph_release_buffers", "gn2v_graph_walk_accel", "gn2v_stats_reset", "gn2v_stats_read",
]
# include/gn2v_internal.h: the steps gn2v_train_blocks is made of (the multi-process trainer and
# the parity tests drive them one at a time) and the measurement utilities
INTERNAL_EXPORTS = [
    "gn2v_block_plan_check", "gn2v_init_table_rows", "gn2v_block_alias_temp_bytes",
    "gn2v_block_alias", "gn2v_block_placement_temp_bytes", "gn2v_block_placement",
    "gn2v_block_place_walks", "gn2v_block_count", "gn2v_block_extract_temp_bytes",
    "gn2v_block_extract", "gn2v_block_cell_offsets", "gn2v_block_step", "gn2v_block_round",
    "gn2v_block_auto_plan", "gn2v_block_auto_plan_graph", "gn2v_block_round_plan",
    "gn2v_graph_xcds", "gn2v_graph_reserve_cus", "gn2v_touch_rows",
]
EXPORTS = BOUNDARY_EXPORTS + INTERNAL_EXPORTS
# include/gn2v_rccl.h: a gn2v_comm filled with RCCL calls, for hosts without Python (RCCL is
# dlopen'ed on first use; this package hands torch.distributed's communicator over instead)
RCCL_EXPORTS = ["gn2v_rccl_unique_id", "gn2v_rccl_comm_create", "gn2v_rccl_comm_destroy"]


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


class GloveIO(C.Structure):
    _fields_ = [(name, C.c_void_p) for name in (
        "d_rows", "d_cols", "d_logx", "d_fx", "d_central", "d_contextual", "d_bias_central",
        "d_bias_contextual")]


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


class StepIO(C.Structure):
    _fields_ = [
        ("d_walks", C.c_void_p),
        ("d_walk_rows", C.c_void_p),
        ("d_central", C.c_void_p),
        ("d_contextual", C.c_void_p),
        ("d_negative", C.c_void_p),
        ("d_neg_pool", C.c_void_p),
        ("neg_pool_size", C.c_uint64),
        ("neg_id_mul", C.c_uint32),
        ("neg_id_add", C.c_uint32),
        ("d_neg_override", C.c_void_p),
        ("d_context_delta", C.c_void_p),
    ]


class BlockPlan(C.Structure):
    """gn2v_block_plan: how nodes are striped over ranks, context parts and XCD slices."""

    _fields_ = [(name, C.c_uint32) for name in (
        "world", "rank", "parts", "slices", "walk_length", "window", "min_dist", "record",
        "row_bits", "flags", "hot_rows", "hot_flush", "key_bits", "ctx_bits")]


class BlockIO(C.Structure):
    _fields_ = [
        ("d_pairs", C.c_void_p),
        ("d_cell_offsets", C.c_void_p),
        ("d_alias", C.c_void_p),
        ("d_cell_rows", C.c_void_p),
        ("d_hot_list", C.c_void_p),
        ("d_hot_slot", C.c_void_p),
        ("d_central", C.c_void_p),
        ("d_context", C.c_void_p),
        ("block_id", C.c_uint64),
        ("part", C.c_uint32),
        ("central_ld", C.c_uint64),
        ("context_ld", C.c_uint64),
        ("d_inv", C.c_void_p),
        ("d_context_table", C.c_void_p),
    ]


class BlockRoundIO(C.Structure):
    """gn2v_block_round_io (include/gn2v_internal.h)."""
    _fields_ = [
        ("d_walks", C.c_void_p),
        ("d_placed_walks", C.c_void_p),
        ("d_inv", C.c_void_p),
        ("d_context_table", C.c_void_p),
        ("d_alias", C.c_void_p),
        ("d_cell_rows", C.c_void_p),
        ("d_hub_bits", C.c_void_p),
        ("d_hot_list", C.c_void_p),
        ("d_hot_slot", C.c_void_p),
        ("d_central", C.c_void_p),
        ("context_parts", C.POINTER(C.c_void_p)),
        ("context_ld", C.c_uint64),
        ("d_work", C.c_void_p),
        ("d_cell_offsets", C.c_void_p),
        ("d_pairs", C.c_void_p),
        ("pairs_capacity", C.c_uint64),
        ("d_temp", C.c_void_p),
        ("temp_bytes", C.c_uint64),
        ("group_parts", C.c_uint32),
        ("next_unit", C.c_uint32),
        ("needed_pairs", C.c_uint64),
        ("pairs_trained", C.c_uint64),
        ("d_pairs2", C.c_void_p),
        ("d_cell_offsets2", C.c_void_p),
        ("d_work2", C.c_void_p),
        ("train_after", C.c_void_p),  # optional hipEvent_t the training launches wait for
    ]


ROUND_GROW = 3  # GN2V_ROUND_GROW

# gn2v_comm (include/gn2v.h): the communicator a host fills for gn2v_train_world
COMM_ALL_GATHER = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p)
COMM_SENDRECV_START = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint32,
                                  C.c_void_p, C.c_uint64, C.c_uint32, C.c_void_p,
                                  C.POINTER(C.c_void_p))
COMM_SENDRECV_WAIT = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p)
COMM_BROADCAST = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint32, C.c_void_p)


class Comm(C.Structure):
    _fields_ = [
        ("ctx", C.c_void_p),
        ("rank", C.c_uint32),
        ("world", C.c_uint32),
        ("all_gather", COMM_ALL_GATHER),
        ("sendrecv_start", COMM_SENDRECV_START),
        ("sendrecv_wait", COMM_SENDRECV_WAIT),
        ("broadcast", COMM_BROADCAST),
    ]


# include/gn2v_experimental.h: measured-and-rejected designs kept for their scripts and tests
EXPERIMENTAL_EXPORTS = ["gn2v_step"]

BLOCK_WORK_WORDS = 532480   # GN2V_BLOCK_WORK_WORDS
BLOCK_MAX_GROUP_CELLS = 16384  # GN2V_BLOCK_MAX_GROUP_CELLS
BLOCK_MAX_WIDE_GROUP_CELLS = 65536  # GN2V_BLOCK_MAX_WIDE_GROUP_CELLS
BLOCK_HOT_MAX = 192      # GN2V_BLOCK_HOT_MAX
BLOCK_HOT_DEFAULT = 192  # GN2V_BLOCK_HOT_DEFAULT


class Stats(C.Structure):
    _fields_ = [
        ("pairs", C.c_uint64),
        ("walk_steps", C.c_uint64),
        ("centres", C.c_uint64),
        ("train_ms", C.c_double),
        ("walk_ms", C.c_double),
        ("train_launches", C.c_uint32),
        ("walk_launches", C.c_uint32),
        ("block_parts", C.c_uint32),
        ("block_slices", C.c_uint32),
        ("block_stripes", C.c_uint32),
        ("block_group_parts", C.c_uint32),
        ("block_round_walks", C.c_uint64),
        ("resident_launches", C.c_uint32),
        ("resident_record", C.c_uint32),
    ]

    def as_dict(self):
        return {name: getattr(self, name) for name, _ in self._fields_}


class Gn2vError(RuntimeError):
    """An entry point of libgn2v.so returned a non-zero status."""


def build(force: bool = False, verbose: bool = False) -> str:
    """Compile ``libgn2v.so`` for gfx950 in-tree with hipcc (cross-compiles without a GPU)."""
    import glob

    # every header and unit in csrc/ is a dependency: a stale library must never survive an edit
    srcs = sorted(glob.glob(os.path.join(_CSRC, "*.h")) + glob.glob(os.path.join(_CSRC, "*.hip")))
    srcs += [_HEADER, _HEADER_EXPERIMENTAL, _HEADER_INTERNAL, _HEADER_RCCL]
    if not force and os.path.exists(LIB_PATH):
        newest = max(os.path.getmtime(s) for s in srcs)
        if os.path.getmtime(LIB_PATH) >= newest:
            return LIB_PATH
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    cmd = [
        hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-munsafe-fp-atomics", "-shared",
        "-fPIC", "-parallel-jobs=2", *os.environ.get("GN2V_HIPCC_FLAGS", "").split(),
        *[os.path.join(_CSRC, u) for u in _UNITS], "-o", LIB_PATH,
    ]
    res = subprocess.run(cmd, capture_output=True, text=True)
    if res.returncode != 0:
        raise RuntimeError("building libgn2v.so failed:\n" + res.stderr)
    if verbose and res.stderr:
        print(res.stderr)
    return LIB_PATH


_lib = None


def lib():
    """Load the shared object; raises if it has not been built (no fallback)."""
    global _lib
    if _lib is not None:
        return _lib
    try:
        # PyTorch's ROCm wheel bundles its own HIP runtime.  It must be the first HIP runtime in
        # the process: loading libgn2v.so first would bind the system libamdhip64 and leave torch
        # unable to see the GPU (two runtimes in one process).
        import torch  # noqa: F401
    except ImportError:
        pass
    if not os.path.exists(LIB_PATH):
        raise ModuleNotFoundError(
            f"The gn2v HIP engine `{LIB_PATH}` has not been built. Run "
            "`python -c 'import __graft_entry__ as g; g.build()'` (needs hipcc from ROCm); "
            "there is deliberately no CPU fallback for this path."
        )
    L = C.CDLL(LIB_PATH)
    vp, u64, u32, f32, i32 = C.c_void_p, C.c_uint64, C.c_uint32, C.c_float, C.c_int
    L.gn2v_version.restype = i32
    L.gn2v_last_error.restype = C.c_char_p
    L.gn2v_device_count.restype = i32
    L.gn2v_graph_create.argtypes = [vp, vp, vp, vp, u64, u64, u64, u32, i32, C.POINTER(vp)]
    L.gn2v_graph_destroy.argtypes = [vp]
    L.gn2v_graph_set_types.argtypes = [vp, vp, vp]
    L.gn2v_ba_edges.argtypes = [u64, u32, u64, vp, vp, vp]
    L.gn2v_walks.argtypes = [vp, C.POINTER(WalkParams), u64, u64, u64, u64, vp, vp]
    L.gn2v_walks_strided.argtypes = [vp, C.POINTER(WalkParams), u64, u64, u64, u64, u32, u64, vp, vp]
    L.gn2v_window_batch.argtypes = [vp, u64, u32, u32, vp, vp, vp]
    L.gn2v_walk_pairs.argtypes = [vp, u64, u32, u32, u32, vp, vp]
    L.gn2v_init_table.argtypes = [vp, u64, u32, u32, u64, u32, f32, vp]
    L.gn2v_cooc_slots.argtypes = [vp, u64, u32, u32, u32, vp, vp, vp]
    L.gn2v_glove_step.argtypes = [vp, C.POINTER(GloveIO), u64, u32, u32, f32, u32, vp]
    step = [vp, C.POINTER(TrainParams), vp, u64, u32, u64, u64, u64, f32, vp, vp, vp, vp]
    L.gn2v_sgns_step.argtypes = step
    L.gn2v_cbow_step.argtypes = step
    L.gn2v_step.argtypes = [vp, C.POINTER(TrainParams), C.POINTER(StepIO), u64, u32, u64, u64,
                            u64, f32, vp]
    L.gn2v_train.argtypes = [vp, C.POINTER(WalkParams), C.POINTER(TrainParams), u64, u64, vp, vp,
                             C.POINTER(Stats), vp]
    L.gn2v_edge_embedding.argtypes = [vp, vp, u32, u32, vp, vp, u64, u32, vp, u32, vp]
    L.gn2v_touch_rows.argtypes = [vp, u32, vp, u64, u32, vp]
    L.gn2v_block_plan_check.argtypes = [vp, C.POINTER(BlockPlan)]
    L.gn2v_init_table_rows.argtypes = [vp, u64, u32, u32, u64, u32, f32, u64, u64, vp]
    L.gn2v_block_alias_temp_bytes.argtypes = [u64, C.POINTER(u64)]
    L.gn2v_block_alias.argtypes = [vp, C.POINTER(BlockPlan), vp, vp, vp, vp, vp, vp, vp, u64, vp]
    L.gn2v_block_placement_temp_bytes.argtypes = [u64, C.POINTER(u64)]
    L.gn2v_block_placement.argtypes = [vp, u32, u64, u64, vp, vp, vp, u64, vp]
    L.gn2v_block_place_walks.argtypes = [vp, vp, u64, vp, vp]
    L.gn2v_block_count.argtypes = [vp, C.POINTER(BlockPlan), vp, vp, u64, u64, u64, u64, u32, u32,
                                   vp, vp, vp]
    L.gn2v_block_extract_temp_bytes.argtypes = [u64, C.POINTER(u64)]
    L.gn2v_block_extract.argtypes = [vp, C.POINTER(BlockPlan), vp, vp, u64, u64, u64, u64, u32,
                                     u32, vp, vp, u64, vp, vp, u64, vp]
    L.gn2v_block_cell_offsets.argtypes = [vp, C.POINTER(BlockPlan), u32, vp, u64, vp, vp]
    L.gn2v_graph_xcds.argtypes = [vp]
    L.gn2v_graph_reserve_cus.argtypes = [vp, u32, vp]
    L.gn2v_block_step.argtypes = [vp, C.POINTER(TrainParams), C.POINTER(BlockPlan),
                                  C.POINTER(BlockIO), u64, u64, f32, vp]
    L.gn2v_block_round.argtypes = [vp, C.POINTER(TrainParams), C.POINTER(BlockPlan), u32,
                                   C.POINTER(BlockRoundIO), u64, u64, u64, u64, f32, u64, vp]
    L.gn2v_block_auto_plan.argtypes = [u64, u32, u32, u32, C.POINTER(u32), C.POINTER(u32)]
    L.gn2v_block_auto_plan_graph.argtypes = [vp, u32, u32, u32, C.POINTER(u32), C.POINTER(u32), vp]
    L.gn2v_block_round_plan.argtypes = [u64, u64, u32, u32, u32, u32, u32, u32, C.POINTER(u64),
                                        C.POINTER(u32)]
    L.gn2v_train_world.argtypes = [vp, C.POINTER(WalkParams), C.POINTER(TrainParams), u64, u64,
                                   u64, C.POINTER(Comm), vp, vp, C.POINTER(Stats), vp]
    L.gn2v_train_blocks.argtypes = [vp, C.POINTER(WalkParams), C.POINTER(TrainParams), u64, u64,
                                    u64, u32, vp, vp, C.POINTER(Stats), vp]
    L.gn2v_graph_release_buffers.argtypes = [vp]
    L.gn2v_graph_walk_accel.argtypes = [vp]
    L.gn2v_stats_reset.argtypes = [vp, vp]
    L.gn2v_stats_read.argtypes = [vp, C.POINTER(Stats), vp]
    for name in EXPORTS + EXPERIMENTAL_EXPORTS:
        fn = getattr(L, name)
        if name not in ("gn2v_last_error",):
            fn.restype = i32
    _lib = L
    return L


def check(status: int):
    if status != 0:
        raise Gn2vError(lib().gn2v_last_error().decode("utf-8", "replace"))


def device_count() -> int:
    return lib().gn2v_device_count()


def require_device():
    """Fail loudly when the engine cannot run (missing library or no GPU)."""
    if device_count() < 1:
        raise RuntimeError(
            "No AMD GPU is visible to HIP. The gn2v engine (embiggen_amd) only runs on a ROCm "
            "device such as MI355X; it has no CPU execution path."
        )

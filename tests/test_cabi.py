"""The C-ABI library loads and exports every symbol include/gn2v.h declares (no compute calls:
this runs without a GPU)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from embiggen_amd import _lib

HEADER = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include",
                      "gn2v.h")


EXPERIMENTAL = os.path.join(os.path.dirname(HEADER), "gn2v_experimental.h")
INTERNAL = os.path.join(os.path.dirname(HEADER), "gn2v_internal.h")
RCCL = os.path.join(os.path.dirname(HEADER), "gn2v_rccl.h")


def declared_symbols(header=HEADER):
    text = open(header).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(gn2v_[a-z_0-9]+)\s*\(", text)))


def test_header_and_binding_agree():
    # the drop-in boundary: what a binding of node2vec.py:99 needs -- two dozen entry points
    assert declared_symbols() == sorted(_lib.BOUNDARY_EXPORTS) and len(_lib.BOUNDARY_EXPORTS) <= 24
    # the steps the block fit is made of: their own header, outside the boundary
    assert declared_symbols(INTERNAL) == sorted(_lib.INTERNAL_EXPORTS)
    assert not set(_lib.INTERNAL_EXPORTS) & set(_lib.BOUNDARY_EXPORTS)
    # rejected designs live in a header of their own, outside the drop-in boundary
    assert declared_symbols(EXPERIMENTAL) == sorted(_lib.EXPERIMENTAL_EXPORTS)
    assert not set(_lib.EXPERIMENTAL_EXPORTS) & set(_lib.EXPORTS)
    # the RCCL-filled communicator of gn2v_train_world, for hosts without Python
    assert declared_symbols(RCCL) == sorted(_lib.RCCL_EXPORTS)
    # nothing else lives in include/
    assert sorted(os.listdir(os.path.dirname(HEADER))) == [
        "gn2v.h", "gn2v_experimental.h", "gn2v_internal.h", "gn2v_rccl.h"]


def test_library_exports_every_declared_symbol():
    L = C.CDLL(_lib.build())
    for name in (declared_symbols() + declared_symbols(INTERNAL) + declared_symbols(EXPERIMENTAL)
                 + declared_symbols(RCCL)):
        assert hasattr(L, name), name
    assert _lib.lib().gn2v_version() == 320


def test_rccl_communicator_refuses_bad_arguments_without_a_gpu():
    """include/gn2v_rccl.h: RCCL is loaded on first use, not linked -- the library loads and
    these calls return 1 with a message where there is no RCCL device (or no RCCL at all)."""
    L = C.CDLL(_lib.build())
    L.gn2v_last_error.restype = C.c_char_p
    comm = _lib.Comm()
    assert L.gn2v_rccl_comm_create(None, 0, 1, 0, C.byref(comm)) == 1
    assert b"NULL" in L.gn2v_last_error()
    ident = (C.c_ubyte * 128)()
    assert L.gn2v_rccl_comm_create(ident, 3, 2, 0, C.byref(comm)) == 1
    assert b"rank < world" in L.gn2v_last_error()
    assert L.gn2v_rccl_comm_destroy(C.byref(comm)) == 0  # nothing to release
    import subprocess
    deps = subprocess.run(["ldd", _lib.build()], capture_output=True, text=True).stdout
    assert "rccl" not in deps and "nccl" not in deps, deps


def test_struct_layouts_match_header():
    assert C.sizeof(_lib.WalkParams) == 32
    assert C.sizeof(_lib.TrainParams) == 48
    assert C.sizeof(_lib.Stats) == 80
    assert C.sizeof(_lib.BlockPlan) == 14 * 4 and C.sizeof(_lib.BlockIO) == 112
    text = open(HEADER).read()
    for name, value in (("GN2V_TRAIN_SCALE_FREE", _lib.TRAIN_SCALE_FREE),
                        ("GN2V_TRAIN_DOWNSAMPLE", _lib.TRAIN_DOWNSAMPLE),
                        ("GN2V_TRAIN_NORM_LR", _lib.TRAIN_NORM_LR),
                        ("GN2V_TRAIN_DETERMINISTIC", _lib.TRAIN_DETERMINISTIC),
                        ("GN2V_TRAIN_ATOMIC", _lib.TRAIN_ATOMIC),
                        ("GN2V_TRAIN_WRITE_BACK", _lib.TRAIN_WRITE_BACK),
                        ("GN2V_TRAIN_WRITE_THROUGH", _lib.TRAIN_WRITE_THROUGH),
                        ("GN2V_TRAIN_WALK_ORDERED", _lib.TRAIN_WALK_ORDERED),
                        ("GN2V_TRAIN_BLOCK_PATH", _lib.TRAIN_BLOCK_PATH),
                        ("GN2V_TRAIN_CENTRAL_STORE", _lib.TRAIN_CENTRAL_STORE),
                        ("GN2V_GRAPH_DEVICE_PTRS", _lib.GRAPH_DEVICE_PTRS),
                        ("GN2V_GRAPH_SYMMETRIC", _lib.GRAPH_SYMMETRIC)):
        assert re.search(rf"#define {name} +{value}u", text), name


def test_automatic_plan_of_the_block_path():
    """gn2v_block_auto_plan (pure host function: no GPU needed): one slice per XCD (8) or none --
    never 2 or 4, which would have several XCDs share a slice --, cells of >= 32 768 rows once
    there are several parts, any number of parts (a multiple of the ranks, at least two per
    rank)."""
    from embiggen_amd.distributed import auto_plan

    assert auto_plan(34, 1) == (1, 1) and auto_plan(34, 8) == (16, 1)
    # one GPU: a single part in 8 slices from GN2V_BLOCK_PATH_MIN_NODES nodes up (where gn2v_train
    # starts to use the block path); travelling parts keep cells of >= 8 192 rows
    from embiggen_amd import _lib
    assert _lib.BLOCK_PATH_MIN_NODES == 2560
    assert auto_plan(2_559, 1) == (1, 1) and auto_plan(2_560, 1) == (1, 8)
    assert auto_plan(2_708, 1) == (1, 8) and auto_plan(40_000, 1) == (1, 8)
    assert auto_plan(40_000, 2) == (4, 1)
    assert auto_plan(65_536, 1) == (1, 8) and auto_plan(169_343, 1) == (1, 8)
    assert auto_plan(262_144, 1) == (1, 8) and auto_plan(1_000_000, 1) == (3, 8)
    assert auto_plan(2_449_029, 1) == (9, 8) and auto_plan(2_449_029, 8) == (16, 8)
    assert auto_plan(200_000, 8) == (16, 1)
    assert [auto_plan(10_000_000, w) for w in (1, 2, 4, 8)] == [(38, 8), (38, 8), (36, 8), (32, 8)]
    assert [auto_plan(100_000_000, w) for w in (1, 8)] == [(381, 8), (376, 8)]
    for n in (10 ** 5, 10 ** 6, 10 ** 7, 10 ** 8, 3 * 10 ** 9):
        for world in (1, 2, 3, 8):
            parts, slices = auto_plan(n, world)
            assert parts % world == 0 and parts * slices <= 524288 and slices in (1, 8)
            assert world == 1 or parts >= 2 * world
            if parts > (1 if world == 1 else 2 * world) and parts * slices < 524288 - 8 * world:
                assert 32768 <= n // (parts * slices) < 2 * 32768 * (world + 1)
    # one GPU, the row width known (ld <= 128 floats): RESIDENT CELLS -- every cell fits one
    # workgroup's LDS (160 KB minus the sixteen waves' staging) -- up to GN2V_RESIDENT_MAX_NODES
    # (115 M: 524 288 cells of 220 rows at d = 128)
    def rows(n, parts, slices):
        return -(-(-(-n // parts)) // slices)  # the largest cell: ceil(ceil(n / parts) / slices)

    assert auto_plan(2_559, 1, 128, 10) == (1, 1)  # below the block path's limit: atomics
    assert auto_plan(2_708, 1, 128, 10) == (1, 8)        # below GN2V_RESIDENT_MIN_NODES: XCD cells
    assert auto_plan(99_999, 1, 128, 10) == (1, 8) and auto_plan(100_000, 1, 128, 10) == (2, 256)
    assert auto_plan(169_343, 1, 128, 10) == (4, 256)    # config 3's shape: cells of 166 rows
    assert auto_plan(2_449_029, 1, 128, 10) == (44, 256)  # config 4's shape
    assert auto_plan(10_000_000, 1, 128, 10) == (178, 256)  # the bench graph: cells of 220 rows
    assert auto_plan(100_000_000, 1, 128, 10) == (1776, 256)  # config 5: 454 656 cells
    assert auto_plan(116_000_000, 1, 128, 10) == (442, 8)  # beyond GN2V_RESIDENT_MAX_NODES: XCD cells
    # rows wider than 128 floats: workgroups of eight waves (half the staging) -- cells of 134 rows
    # at 256 floats, of 66 at 512; wider rows keep the XCD cells
    assert auto_plan(1_000_000, 1, 256, 10) == (30, 256)
    assert auto_plan(1_000_000, 1, 512, 10) == (60, 256)
    assert auto_plan(1_000_000, 1, 516, 10) == (3, 8)
    assert auto_plan(1_000_000, 1, 128, 50) == (28, 256)  # 50 negatives: cells of 140 rows
    # several ranks: the parts travel -- two per rank (more only beyond 8 192 slices), each launched
    # by itself with all its cells -- while a part keeps 64 cells; smaller graphs travel as XCD cells
    assert auto_plan(1_000_000, 2, 128, 10) == (4, 1137) and auto_plan(1_000_000, 8, 128, 10) == (16, 285)
    assert auto_plan(10_000_000, 8, 128, 10) == (16, 2841) and auto_plan(10_000_000, 2, 128, 10) == (6, 7576)
    assert auto_plan(200_000, 8, 128, 10) == auto_plan(200_000, 8) == (16, 1)
    for n, world in ((1_000_000, 3), (2_449_029, 8), (10_000_000, 4), (13_000_000, 8), (100_000_000, 8)):
        parts, slices = auto_plan(n, world, 128, 10)
        assert parts % world == 0 and parts >= 2 * world and 64 <= slices <= 8192
        assert rows(n, parts, slices) <= 220 and parts * slices <= 524288
    for n in (100_000, 250_000, 1_000_000, 1_500_000, 3_000_000, 12_999_999, 30_000_000, 300_000_000):
        for ld, k in ((128, 10), (64, 5), (32, 10), (128, 40), (256, 10), (224, 5), (384, 10),
                      (512, 5), (96, 10), (160, 10)):
            parts, slices = auto_plan(n, 1, ld, k)
            # records of 32 pairs, or of 16 / 8 when their staging would leave under 64 rows
            for record in (32, 16, 8):
                # per wave: four transposition rows (128 floats at most), the centres and the
                # 16-bit sample lists ((k + 1) padded to 4, record + 1 of them); shared: the dummy
                # row; per row: the row, its alias entry, its node id
                stride = (k + 1 + 3) // 4 * 4
                words = (4 * min(ld, 128) + record + (record + 1) * stride // 2 + 2 + 3) // 4 * 4
                # (in LDS a row wider than 64 floats is padded to 128, 256 or 512 floats)
                lds_ld = ld if ld <= 64 else 128 if ld <= 128 else 256 if ld <= 256 else 512
                staging = (16 if ld <= 128 else 8) * 4 * words + 64 + lds_ld * 4
                fit = min(4095, max(0, 160 * 1024 - staging) // (lds_ld * 4 + 12))
                if fit >= 64:
                    break
            if slices > 8:
                assert parts * slices <= 524288 and slices <= 256
                assert rows(n, parts, slices) <= fit, (n, ld, k, parts, slices)
            else:
                assert (parts, slices) == auto_plan(n, 1)  # did not fit: the XCD plan


def test_round_size_and_groups_follow_the_free_memory():
    """gn2v_block_round_plan (pure host function): rounds long enough for 64 pairs per (cell,
    centre) inside [2^20, 2^23] walks; per group of parts the pair words (8 B) are held once
    sorted (twice with a group in preparation) and once unsorted, and the walks beside them must
    fit three quarters of the free HBM; at least four groups per round."""
    from embiggen_amd.distributed import round_plan

    GB = 10 ** 9
    # the bench graph on one GPU: 38 x 8 cells, rounds at the cap, four groups of <= 10 parts
    assert round_plan(270 * GB, 10_000_000, 128, 5, 1, 38, 8, False) == (1 << 23, 10)
    # the same graph in resident cells (178 x 256): as few groups as a third of the memory (what
    # a handle keeps between fits) allows -- three at rounds of 2^23 walks, ONE at the rounds the
    # rounds-per-epoch rule trains (below)
    assert round_plan(270 * GB, 10_000_000, 128, 5, 1, 178, 256, False) == (1 << 23, 60)
    # (a wide group: more cells than LDS counters -- 7 groups a round where 13 824 counters gave 33)
    assert round_plan(150 * GB, 100_000_000, 128, 5, 1, 1776, 256, False) == (1 << 23, 254)
    # a group is one launch: at least 4 096 cells when the plan has them, the round shortened (not
    # the group) until the pair words fit a third of the memory -- 1 M nodes: the whole round of
    # 2^21 walks in one launch of 4 608 cells; 2.4 M: two groups of 22 parts, rounds of 2^22
    assert round_plan(270 * GB, 1_000_000, 128, 5, 1, 18, 256, True) == (1 << 21, 18)
    assert round_plan(270 * GB, 2_449_029, 128, 5, 1, 44, 256, True) == (1 << 22, 22)
    assert round_plan(270 * GB, 169_343, 128, 5, 1, 4, 256, True) == (1 << 21, 4)
    # several ranks: every scan of a group reads the walks of ALL ranks -- ONE group a round when
    # memory allows (wide groups: their cell offsets follow the sort), more when it does not
    # (equal groups: 6 + 6 + 4, never 7 + 7 + 2)
    assert round_plan(300 * GB, 10_000_000, 128, 5, 8, 16, 2841, True) == (1 << 23, 8)
    assert round_plan(250 * GB, 10_000_000, 128, 5, 8, 16, 2841, True) == (1 << 23, 6)
    assert round_plan(150 * GB, 10_000_000, 128, 5, 8, 16, 2841, True) == (1 << 23, 2)
    # the plan is made for the round that will be trained (cap: the rounds-per-epoch rule of
    # resident cells, a caller's round): a rank of 8 holds its round of 2^19 walks in one group
    assert round_plan(300 * GB, 10_000_000, 128, 5, 8, 16, 2841, True, cap=1 << 19) == (1 << 19, 16)
    assert round_plan(150 * GB, 10_000_000, 128, 5, 8, 16, 2841, True, cap=1 << 19) == (1 << 19, 16)
    assert round_plan(270 * GB, 10_000_000, 128, 5, 1, 178, 256, True, cap=1_562_500) == (1_562_500, 178)
    assert round_plan(270 * GB, 2_449_029, 128, 5, 1, 44, 256, True, cap=382_661) == (382_661, 44)
    assert round_plan(270 * GB, 169_343, 128, 5, 1, 4, 256, True, cap=26_460) == (26_460, 4)
    assert round_plan(270 * GB, 10_000_000, 128, 5, 1, 178, 256, True, cap=1 << 25) == (1 << 23, 45)
    # (longer walks: four to six times the pairs per walk -- the groups above the floor of 16
    # parts are cut to what a handle keeps, then the round is shortened)
    assert round_plan(270 * GB, 10_000_000, 512, 5, 1, 178, 256, False) == (8388608, 20)
    assert round_plan(270 * GB, 10_000_000, 800, 5, 1, 178, 256, False) == (4194304, 26)
    # eight GPUs, a group in preparation while one trains
    assert round_plan(270 * GB, 10_000_000, 128, 5, 8, 32, 8, True) == (1 << 23, 8)
    # 100 M nodes on one GPU, 170 GB free beside the tables
    assert round_plan(170 * GB, 100_000_000, 128, 5, 1, 381, 8, False) == (1 << 23, 96)
    # a mid-size graph wants no more than it needs; a small one the floor
    assert round_plan(270 * GB, 1_000_000, 128, 5, 1, 3, 8, False) == (1 << 21, 1)
    assert round_plan(270 * GB, 169_343, 128, 5, 1, 1, 8, False) == (1 << 20, 1)
    # short memory: smaller groups first, then shorter rounds
    walks, group = round_plan(20 * GB, 10_000_000, 128, 5, 1, 38, 8, False)
    assert walks == 1 << 23 and group < 10
    walks, group = round_plan(2 * GB, 10_000_000, 128, 5, 1, 38, 8, False)
    assert walks < 1 << 23 and group == 1
    assert round_plan(0, 10_000_000, 128, 5, 1, 38, 8, False) == (1 << 14, 1)
    for free in (GB, 10 * GB, 100 * GB, 10 ** 13):
        for world, overlap, parts in ((1, False, 38), (2, True, 38), (8, True, 32)):
            n, gp = round_plan(free, 10_000_000, 128, 5, world, parts, 8, overlap)
            assert n & (n - 1) == 0 and (1 << 14) <= n <= (1 << 23) and 1 <= gp <= (parts + 3) // 4
            need = (4 * 128 * n * (world + 1 if world > 1 else 1)
                    + (3 if overlap else 2) * 8 * (n * 1280 // parts) * gp)
            assert (n == 1 << 14 and gp == 1) or need <= 0.75 * free
    with pytest.raises(RuntimeError):
        round_plan(GB, 10_000_000, 1, 5, 1, 38, 8, False)


def test_errors_are_reported_not_thrown():
    L = _lib.lib()
    handle = C.c_void_p()
    rp = np.array([0, 1, 2], dtype=np.uint64)
    ci = np.array([1, 0], dtype=np.uint32)
    assert L.gn2v_graph_create(None, None, None, None, 2, 2, 2, 0, 0, C.byref(handle)) != 0
    assert b"NULL" in L.gn2v_last_error()
    assert L.gn2v_graph_create(rp.ctypes.data, ci.ctypes.data, None, None, 0, 2, 0, 0, 0,
                               C.byref(handle)) != 0
    assert b"no nodes" in L.gn2v_last_error()
    if _lib.device_count() == 0:
        assert L.gn2v_graph_create(rp.ctypes.data, ci.ctypes.data, None, None, 2, 2, 2, 0, 0,
                                   C.byref(handle)) != 0
        assert b"no CPU fallback" in L.gn2v_last_error()
        with pytest.raises(RuntimeError):
            _lib.require_device()
    assert L.gn2v_walks(None, None, 0, 0, 0, 0, None, None) != 0
    assert L.gn2v_train(None, None, None, 0, 0, None, None, None, None) != 0
    assert L.gn2v_graph_destroy(None) == 0


def test_header_is_valid_c_and_a_plain_c_program_can_call_the_library(tmp_path):
    """What a cgo / Rust-FFI / JNI binding relies on: include/gn2v.h compiles as strict C11 (and as
    C++17), and a C program linked against libgn2v.so (tests/c/abi_smoke.c) gets the host-only
    entry points and the error convention to work -- no GPU, no Python in the process."""
    import shutil
    import subprocess

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    header = os.path.join(root, "include", "gn2v.h")
    if shutil.which("gcc") is None:
        pytest.skip("no C compiler")
    for h in (header, INTERNAL):
        subprocess.run(["gcc", "-std=c11", "-Wall", "-Wextra", "-Werror", "-pedantic",
                        "-fsyntax-only", "-x", "c", h], check=True)
        subprocess.run(["g++", "-std=c++17", "-Wall", "-Werror", "-fsyntax-only", "-x", "c++", h],
                       check=True)
    lib_dir = os.path.dirname(_lib.build())
    exe = str(tmp_path / "abi_smoke")
    subprocess.run(["gcc", "-std=c11", "-Wall", "-Wextra", "-Werror", "-pedantic",
                    "-I", os.path.join(root, "include"), os.path.join(root, "tests", "c", "abi_smoke.c"),
                    "-L", lib_dir, "-lgn2v", f"-Wl,-rpath,{lib_dir}", "-o", exe], check=True)
    res = subprocess.run([exe], capture_output=True, text=True)
    assert res.returncode == 0, (res.returncode, res.stdout, res.stderr)
    assert f"sizeof(gn2v_stats) = {C.sizeof(_lib.Stats)}" in res.stdout
    assert f"sizeof(gn2v_block_plan) = {C.sizeof(_lib.BlockPlan)}" in res.stdout
    assert f"sizeof(gn2v_block_io) = {C.sizeof(_lib.BlockIO)}" in res.stdout

"""The CPU twins of the boundary's compute entry points (oracle/gn2v_cpu.h, SURVEY.md 8b "CPU twins
gn2v_cpu_* with the same signatures"): they are exported by the ORACLE's library only, take their
namesakes' argument lists, and are the oracle (so everything the oracle is pinned on pins them)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import embiggen_amd as E
from embiggen_amd import _lib
from oracle import oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _prototypes(path, prefix):
    """{name: [parameter types]} of the `int <prefix>...(...)` declarations of a header."""
    text = re.sub(r"/\*.*?\*/", " ", open(path).read(), flags=re.S)
    out = {}
    for m in re.finditer(r"\bint\s+(%s\w+)\s*\(([^;{]*?)\)\s*;" % prefix, text):
        params = []
        for p in m.group(2).split(","):
            p = " ".join(p.split())
            if p in ("void", ""):
                continue
            # drop the parameter's name, keep its type (pointers stay with the type)
            params.append(re.sub(r"\s*\b\w+$", "", p.replace("*", " * ")).replace(" ", ""))
        out[m.group(1)] = params
    return out


def test_twins_have_their_namesakes_signatures():
    twins = _prototypes(os.path.join(ROOT, "oracle", "gn2v_cpu.h"), "gn2v_cpu_")
    boundary = _prototypes(os.path.join(ROOT, "include", "gn2v.h"), "gn2v_")
    assert len(twins) >= 9
    for name, params in twins.items():
        original = "gn2v_" + name[len("gn2v_cpu_"):]
        assert original in boundary, f"{name} has no namesake in include/gn2v.h"
        want = [t.replace("gn2v_graph", "gn2v_cpu_graph") for t in boundary[original]]
        assert params == want, (name, params, want)


def test_twins_live_in_the_oracle_library_only():
    twins = _prototypes(os.path.join(ROOT, "oracle", "gn2v_cpu.h"), "gn2v_cpu_")
    L = O.lib()
    for name in list(twins) + ["gn2v_cpu_last_error"]:
        assert hasattr(L, name), name
    product = C.CDLL(_lib.build())
    for name in twins:
        assert not hasattr(product, name), f"libgn2v.so must not carry a CPU path ({name})"


@pytest.fixture(scope="module")
def graphs():
    src, dst = O.ba_edges(300, 3, 5)
    g = E.CSRGraph.from_edge_list(src, dst, number_of_nodes=300)
    return g, O.OracleGraph(g.row_ptr, g.col_idx), O.CpuGraph(g.row_ptr, g.col_idx)


def test_walks_window_batch_and_tables_are_the_oracles(graphs):
    g, og, cg = graphs
    # the device library's own parameter struct goes straight into the twin
    wp = _lib.WalkParams(24, 2, 0.5, 2.0, 100, 0, 0.0, 0.0)
    owp = O.WalkParams(24, 2, 0.5, 2.0, 100, 0)
    got = O.cpu_walks(cg, wp, 9, 1, 17, 200)
    assert np.array_equal(got, O.walks(og, owp, 9, 1, 17, 200))
    # ids in groups: 3 groups of 70 walks (the last one short), 1 000 ids apart
    strided = O.cpu_walks_strided(cg, wp, 9, 1, 17, 200, 70, 1000)
    assert np.array_equal(strided, np.concatenate(
        [O.walks(og, owp, 9, 1, 17 + q * 1000, min(70, 200 - 70 * q)) for q in range(3)]))
    ctx, words = O.cpu_window_batch(got, 3)
    want_ctx, want_words = O.window_batch(got, 3)
    assert np.array_equal(ctx, want_ctx) and np.array_equal(words, want_words)
    assert ctx.shape == (200 * (24 - 6), 6)
    assert np.array_equal(O.cpu_init_table(50, 7, 8, 3, 1, 0.25), O.init_table(50, 7, 8, 3, 1, 0.25))


@pytest.mark.parametrize("model", [0, 1])
def test_steps_and_fit_are_the_oracles(graphs, model):
    g, og, cg = graphs
    n, d, ld = g.get_number_of_nodes(), 12, 12
    wp = _lib.WalkParams(16, 2, 0.25, 4.0, 100, 0, 0.0, 0.0)
    # flags beyond the oracle's three (here: the device's DETERMINISTIC bit) are ignored
    tp = _lib.TrainParams(1 - model, d, ld, 2, 4, 3, 0.05, 0.9, 6.0, 1 | _lib.TRAIN_DETERMINISTIC,
                          d ** -0.5, 0)
    otp = O.TrainParams(model, d, ld, 2, 4, 3, 0.05, 0.9, 6.0, 1, d ** -0.5)
    wk = O.cpu_walks(cg, wp, 4, 0, 0, 120)
    c, x = O.cpu_init_table(n, d, ld, 4, 0, d ** -0.5), O.cpu_init_table(n, d, ld, 4, 1, d ** -0.5)
    c2, x2 = c.copy(), x.copy()
    O.cpu_step(cg, tp, model, wk, 4, 0, 0, 0.05, c, x)   # tp.model (the other one) is ignored
    O.train_walks(og, otp, wk, 4, 0, 0, 0.05, c2, x2)
    assert np.array_equal(c, c2) and np.array_equal(x, x2)
    assert np.abs(c - O.init_table(n, d, ld, 4, 0, d ** -0.5)).max() > 0

    tp.model = model
    stats = _lib.Stats()
    fc, fx = O.cpu_train(cg, wp, tp, 11, stats=stats)
    rc, rx, pairs = O.fit(og, O.WalkParams(16, 2, 0.25, 4.0, 100, 0), otp, 11)
    assert np.array_equal(fc, rc) and np.array_equal(fx, rx)
    assert stats.pairs == pairs and stats.walk_steps == 2 * 2 * n * 15
    assert 0 < stats.centres <= 2 * 2 * n * 16
    # the walk budget of gn2v_train: only that many walks an epoch
    stats2 = _lib.Stats()
    O.cpu_train(cg, wp, tp, 11, max_walks_per_epoch=100, stats=stats2)
    assert stats2.walk_steps == 2 * 100 * 15


def test_hogwild_threads_train_the_same_pairs(graphs):
    g, _, _ = graphs
    cg = O.CpuGraph(g.row_ptr, g.col_idx, threads=4)
    wp = _lib.WalkParams(16, 1, 1.0, 1.0, 0, 0, 0.0, 0.0)
    tp = _lib.TrainParams(0, 8, 8, 1, 3, 2, 0.05, 0.9, 6.0, 1, 8 ** -0.5, 0)
    a, b = _lib.Stats(), _lib.Stats()
    c4, _ = O.cpu_train(cg, wp, tp, 3, stats=a)
    c1, _ = O.cpu_train(O.CpuGraph(g.row_ptr, g.col_idx), wp, tp, 3, stats=b)
    assert a.pairs == b.pairs and np.isfinite(c4).all()
    # same walks, same samples; only the order of the updates differs
    cos = float((c4 * c1).sum() / np.sqrt((c4 * c4).sum() * (c1 * c1).sum()))
    assert cos > 0.98, cos


def test_bad_arguments_return_one_with_a_message(graphs):
    _, _, cg = graphs
    L = O.lib()
    wp = _lib.WalkParams(1, 1, 1.0, 1.0, 0, 0, 0.0, 0.0)  # walk_length 1
    out = np.zeros((1, 1), dtype=np.uint32)
    assert L.gn2v_cpu_walks(cg.handle, C.byref(wp), C.c_uint64(0), C.c_uint64(0), C.c_uint64(0),
                            C.c_uint64(1), out.ctypes.data_as(C.c_void_p), None) == 1
    assert b"walk_length" in L.gn2v_cpu_last_error()
    assert L.gn2v_cpu_walks(None, C.byref(wp), C.c_uint64(0), C.c_uint64(0), C.c_uint64(0),
                            C.c_uint64(1), out.ctypes.data_as(C.c_void_p), None) == 1
    handle = C.c_void_p()
    rp = np.array([0, 1, 3], dtype=np.uint64)  # row_ptr[n] != n_edges
    ci = np.array([1, 0], dtype=np.uint32)
    assert L.gn2v_cpu_graph_create(rp.ctypes.data_as(C.c_void_p), ci.ctypes.data_as(C.c_void_p),
                                   None, None, C.c_uint64(2), C.c_uint64(2), C.c_uint64(0),
                                   C.c_uint32(0), C.c_int(1), C.byref(handle)) == 1
    with pytest.raises(O.CpuTwinError):
        O.cpu_check(1)

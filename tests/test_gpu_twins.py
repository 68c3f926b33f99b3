"""The boundary's compute entry points against their CPU twins (oracle/gn2v_cpu.h): the SAME
argument lists -- the same parameter structs, seeds, walk ids -- go into libgn2v.so (device
pointers) and into the oracle's library (host pointers).  Walks and batches bit-exact; tables to
1e-5 after a step, 1e-4 after a fit (f32 sums in another order, v_exp_f32 against libm)."""
import ctypes as C

import numpy as np
import pytest
import torch

import embiggen_amd as E
from embiggen_amd import _lib, ops
from oracle import oracle as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def pair():
    g = E.barabasi_albert(2000, 4, 11)
    return g, O.CpuGraph(g.row_ptr, g.col_idx)


def _call_both(name, g, cg, args_dev, args_host):
    """gn2v_<name>(handle, *args_dev, stream) and gn2v_cpu_<name>(twin, *args_host, NULL)."""
    dg = g.device_graph(0)
    stream = torch.cuda.current_stream().cuda_stream
    _lib.check(getattr(_lib.lib(), "gn2v_" + name)(dg.handle, *args_dev, stream))
    torch.cuda.synchronize()
    O.cpu_check(getattr(O.lib(), "gn2v_cpu_" + name)(cg.handle, *args_host, None))


@pytest.mark.parametrize("max_neighbours", [0, 100, 8])
@pytest.mark.parametrize("weights", [(1.0, 1.0), (0.25, 4.0), (2.0, 0.5)])
def test_walks(pair, weights, max_neighbours):
    g, cg = pair
    wp = _lib.WalkParams(32, 2, weights[0], weights[1], max_neighbours, 0, 0.0, 0.0)
    n = 3000
    dev = torch.empty((n, 32), dtype=torch.int32, device="cuda")
    host = np.empty((n, 32), dtype=np.uint32)
    common = (C.byref(wp), C.c_uint64(5), C.c_uint64(2), C.c_uint64(777), C.c_uint64(n))
    _call_both("walks", g, cg, common + (C.c_void_p(dev.data_ptr()),),
               common + (host.ctypes.data_as(C.c_void_p),))
    assert np.array_equal(dev.cpu().numpy().view(np.uint32), host)


@pytest.mark.parametrize("group,stride,n", [(256, 4000, 4096), (100, 7, 250), (1, 3, 64), (5000, 1, 700)])
def test_walks_strided(pair, group, stride, n):
    """A Node2VecSequence batch in one launch: the same walks as one gn2v_walks call per group,
    and the twin's."""
    g, cg = pair
    wp = _lib.WalkParams(24, 2, 0.5, 2.0, 100, 0, 0.0, 0.0)
    dev = torch.empty((n, 24), dtype=torch.int32, device="cuda")
    host = np.empty((n, 24), dtype=np.uint32)
    common = (C.byref(wp), C.c_uint64(5), C.c_uint64(0), C.c_uint64(31), C.c_uint64(n),
              C.c_uint32(group), C.c_uint64(stride))
    _call_both("walks_strided", g, cg, common + (C.c_void_p(dev.data_ptr()),),
               common + (host.ctypes.data_as(C.c_void_p),))
    assert np.array_equal(dev.cpu().numpy().view(np.uint32), host)
    parts = [ops.walks(g, wp, 5, 0, 31 + q * stride, min(group, n - q * group))
             for q in range(-(-n // group))]
    assert torch.equal(torch.cat(parts), dev)


def test_window_batch_and_init_table(pair):
    g, cg = pair
    wk = O.cpu_walks(cg, _lib.WalkParams(20, 1, 1.0, 1.0, 0, 0, 0.0, 0.0), 1, 0, 0, 500)
    wk_d = torch.from_numpy(wk.view(np.int32)).cuda()
    ctx_d, words_d = ops.window_batch(wk_d, 4)
    ctx, words = O.cpu_window_batch(wk, 4)
    assert np.array_equal(ctx_d.cpu().numpy(), ctx) and np.array_equal(words_d.cpu().numpy(), words)
    for d, ld in ((8, 8), (100, 100), (5, 8)):
        t = ops.init_table(64, d, 9, 1, 0.3, ld=ld)
        assert np.array_equal(t.cpu().numpy(), O.cpu_init_table(64, d, ld, 9, 1, 0.3))


@pytest.mark.parametrize("name,model", [("sgns_step", 0), ("cbow_step", 1)])
@pytest.mark.parametrize("d", [16, 128])
def test_steps(pair, name, model, d):
    g, cg = pair
    n = g.get_number_of_nodes()
    tp = ops.train_params(model, d, 5, 4, flags=1 | _lib.TRAIN_DETERMINISTIC)
    wk = O.cpu_walks(cg, _lib.WalkParams(24, 1, 0.5, 2.0, 100, 0, 0.0, 0.0), 3, 0, 0, 400)
    wk_d = torch.from_numpy(wk.view(np.int32)).cuda()
    c_d, x_d = ops.init_table(n, d, 3, 0, d ** -0.5), ops.init_table(n, d, 3, 1, d ** -0.5)
    c_h, x_h = c_d.cpu().numpy().copy(), x_d.cpu().numpy().copy()
    common = (C.c_uint64(400), C.c_uint32(24), C.c_uint64(3), C.c_uint64(1), C.c_uint64(50),
              C.c_float(0.05))
    _call_both(name, g, cg,
               (C.byref(tp), C.c_void_p(wk_d.data_ptr())) + common
               + (C.c_void_p(c_d.data_ptr()), C.c_void_p(x_d.data_ptr()), None),
               (C.byref(tp), wk.ctypes.data_as(C.c_void_p)) + common
               + (c_h.ctypes.data_as(C.c_void_p), x_h.ctypes.data_as(C.c_void_p), None))
    assert np.abs(c_d.cpu().numpy() - c_h).max() < 1e-5
    assert np.abs(x_d.cpu().numpy() - x_h).max() < 1e-5
    assert np.abs(c_h - O.cpu_init_table(n, d, c_h.shape[1], 3, 0, d ** -0.5)).max() > 1e-4


@pytest.mark.parametrize("model", [0, 1])
def test_train(pair, model):
    """gn2v_train in its reference-semantic, sequential form (walk-ordered, deterministic) is
    gn2v_cpu_train: tables, and the three counters of gn2v_stats."""
    g, cg = pair
    n, d = g.get_number_of_nodes(), 16
    wp = _lib.WalkParams(20, 2, 0.25, 4.0, 100, 0, 0.0, 0.0)
    tp = ops.train_params(model, d, 4, 3, lr=0.02, epochs=2,
                          flags=1 | _lib.TRAIN_DETERMINISTIC | _lib.TRAIN_WALK_ORDERED)
    c_d = torch.empty((n, tp.ld), dtype=torch.float32, device="cuda")
    x_d = torch.empty_like(c_d)
    c_h, x_h = np.empty((n, tp.ld), dtype=np.float32), np.empty((n, tp.ld), dtype=np.float32)
    s_d, s_h = _lib.Stats(), _lib.Stats()
    ops.stats_reset(g, 0)
    common = (C.byref(wp), C.byref(tp), C.c_uint64(21), C.c_uint64(1500))
    _call_both("train", g, cg,
               common + (C.c_void_p(c_d.data_ptr()), C.c_void_p(x_d.data_ptr()), C.byref(s_d)),
               common + (c_h.ctypes.data_as(C.c_void_p), x_h.ctypes.data_as(C.c_void_p),
                         C.byref(s_h)))
    assert (s_d.pairs, s_d.walk_steps, s_d.centres) == (s_h.pairs, s_h.walk_steps, s_h.centres)
    assert s_h.walk_steps == 2 * 1500 * 19 and s_h.pairs > 0
    assert np.abs(c_d.cpu().numpy() - c_h).max() < 1e-4
    assert np.abs(x_d.cpu().numpy() - x_h).max() < 1e-4

"""GPU parity for typed graphs: change_node_type_weight / change_edge_type_weight walks are
bit-identical to the oracle, whose distribution tests/test_typed_walks.py pins."""
import numpy as np
import pytest
import torch

import embiggen_amd as E
from embiggen_amd import _lib, ops
from helpers import typed_karate
from oracle import oracle as O

pytestmark = pytest.mark.gpu

COMBOS = [(1.0, 1.0, 4.0, 1.0), (1.0, 1.0, 1.0, 0.2), (0.25, 4.0, 0.3, 3.0), (2.0, 0.5, 5.0, 5.0),
          (1.0, 1.0, 1e-5, 1.0), (0.5, 2.0, 1.0, 1e-5), (1e-3, 1e-3, 1e-3, 1e-3)]


def _u32(t):
    return t.cpu().numpy().view(np.uint32)


def oracle_graph(g):
    return O.OracleGraph(g.row_ptr, g.col_idx, g.cumw, g.node_type_ids, g.edge_type_ids)


@pytest.mark.parametrize("rw,ew,cn,ce", COMBOS)
@pytest.mark.parametrize("walk_length", [2, 17, 128])
def test_typed_walks_bit_exact_on_karate(rw, ew, cn, ce, walk_length):
    g = typed_karate()
    got = _u32(ops.walks(g, ops.walk_params(walk_length, 10, rw, ew, 100, cn, ce), 42, 1, 0, 340))
    ref = O.walks(oracle_graph(g), O.WalkParams(walk_length, 10, rw, ew, 100, 0, cn, ce), 42, 1, 0, 340)
    assert np.array_equal(got, ref)


def test_type_weights_have_no_impact_without_types(karate, karate_oracle):
    got = _u32(ops.walks(karate, ops.walk_params(30, 2, 0.25, 4.0, 100, 3.0, 0.2), 4, 0, 0, 68))
    assert np.array_equal(got, O.walks(karate_oracle, O.WalkParams(30, 2, 0.25, 4.0, 100, 0), 4, 0, 0, 68))
    # zero-filled fields at the C boundary mean "unset"
    wp = _lib.WalkParams(30, 2, 0.25, 4.0, 100, 0)
    assert np.array_equal(_u32(ops.walks(typed_karate(), wp, 4, 0, 0, 68)), got)


def test_only_one_kind_of_type_attached():
    g = typed_karate()
    nodes_only = g.with_types()  # copy, then strip one kind each
    nodes_only._edge_type_ids = None
    edges_only = g.with_types()
    edges_only._node_type_ids = None
    wp, owp = ops.walk_params(20, 2, 0.5, 2.0, 100, 4.0, 0.25), O.WalkParams(20, 2, 0.5, 2.0, 100, 0, 4.0, 0.25)
    for graph in (nodes_only, edges_only):
        assert np.array_equal(_u32(ops.walks(graph, wp, 6, 0, 0, 68)),
                              O.walks(oracle_graph(graph), owp, 6, 0, 0, 68))


def test_typed_scale_free_multigraph_bit_exact():
    """BA graph with hubs, random node types, and parallel edges of different types."""
    rng = np.random.RandomState(5)
    s, d = O.ba_edges(4000, 4, 3)
    extra = rng.randint(0, len(s), size=3000)  # duplicate some edges under another type
    src = np.concatenate([s, s[extra]])
    dst = np.concatenate([d, d[extra]])
    et = np.concatenate([rng.randint(0, 3, size=len(s)), rng.randint(3, 5, size=len(extra))])
    g = E.CSRGraph.from_edge_list(src, dst, number_of_nodes=4000,
                                  node_types=rng.randint(0, 4, size=4000).tolist(),
                                  edge_types=et.tolist())
    assert g.is_multigraph()
    og = oracle_graph(g)
    for rw, ew, cn, ce in ((0.25, 4.0, 2.0, 0.5), (1.0, 1.0, 0.1, 10.0)):
        got = _u32(ops.walks(g, ops.walk_params(64, 2, rw, ew, 100, cn, ce), 5, 1, 0, 8000))
        assert np.array_equal(got, O.walks(og, O.WalkParams(64, 2, rw, ew, 100, 0, cn, ce), 5, 1, 0, 8000))


def test_weighted_typed_walks_bit_exact():
    rng = np.random.RandomState(3)
    s, d = O.ba_edges(800, 3, 5)
    g = E.CSRGraph.from_edge_list(s, d, rng.uniform(0.1, 5.0, size=len(s)), number_of_nodes=800,
                                  node_types=rng.randint(0, 3, size=800).tolist(),
                                  edge_types=rng.randint(0, 3, size=len(s)).tolist())
    og = oracle_graph(g)
    for rw, ew, cn, ce in ((1.0, 1.0, 3.0, 1.0), (0.25, 4.0, 0.5, 2.0), (1.0, 1e-4, 1e-4, 1e-4)):
        got = _u32(ops.walks(g, ops.walk_params(24, 2, rw, ew, 100, cn, ce), 8, 0, 0, 1600))
        assert np.array_equal(got, O.walks(og, O.WalkParams(24, 2, rw, ew, 100, 0, cn, ce), 8, 0, 0, 1600))


def test_device_resident_graph_with_types():
    """Types attached to a device-built graph (borrowed device pointers)."""
    g = E.barabasi_albert(20000, 5, seed=9)
    dev = torch.device("cuda", 0)
    gen = torch.Generator(device=dev).manual_seed(1)
    nt = torch.randint(0, 3, (g.get_number_of_nodes(),), generator=gen, device=dev, dtype=torch.int32)
    et = torch.randint(0, 4, (g.get_number_of_directed_edges(),), generator=gen, device=dev,
                       dtype=torch.int32)
    tg = g.with_types(nt, et)
    assert tg.has_node_types() and tg.get_number_of_edge_types() == 4
    got = _u32(ops.walks(tg, ops.walk_params(40, 1, 0.25, 4.0, 100, 0.3, 3.0), 2, 0, 0, 20000))
    ref = O.walks(oracle_graph(tg), O.WalkParams(40, 1, 0.25, 4.0, 100, 0, 0.3, 3.0), 2, 0, 0, 20000)
    assert np.array_equal(got, ref)
    # the untyped handle of the same arrays is unaffected
    assert not np.array_equal(got, _u32(ops.walks(g, ops.walk_params(40, 1, 0.25, 4.0, 100, 0.3, 3.0), 2, 0, 0, 20000)))


def test_embedder_and_sequence_use_the_type_weights():
    g = typed_karate()
    kw = dict(embedding_size=16, epochs=2, walk_length=16, iterations=2, window_size=3, verbose=False)
    plain = E.Node2VecSkipGramEnsmallen(**kw).fit_transform(g).get_all_node_embedding()[0].values
    typed = E.Node2VecSkipGramEnsmallen(change_node_type_weight=8.0, change_edge_type_weight=0.1,
                                        **kw).fit_transform(g).get_all_node_embedding()[0].values
    assert np.isfinite(typed).all() and not np.allclose(plain, typed)
    model = E.Node2VecSkipGramEnsmallen(change_node_type_weight=8.0, **kw)
    assert model.is_using_node_types() and not model.is_using_edge_types()
    # on an untyped graph the weights change nothing
    from embiggen_amd import models

    k = E.karate_club()
    kw.pop("verbose")
    a = models.CBOW(deterministic=True, **kw).fit_transform(k)
    b = models.CBOW(deterministic=True, change_node_type_weight=8.0, change_edge_type_weight=0.1,
                    **kw).fit_transform(k)
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
    seq = E.Node2VecSequence(g, walk_length=12, batch_size=34, iterations=2, window_size=2,
                             change_node_type_weight=0.2, change_edge_type_weight=5.0)
    contexts, words = seq[0][0][0]
    assert contexts.shape == (34 * 2 * 8, 4) and words.shape == (34 * 2 * 8,)
    plain_seq = E.Node2VecSequence(g, walk_length=12, batch_size=34, iterations=2, window_size=2)
    assert not np.array_equal(plain_seq[0][0][0][1], words)


def test_set_types_through_the_c_abi():
    import ctypes as C

    g = typed_karate()
    L = _lib.lib()
    handle = C.c_void_p()
    _lib.check(L.gn2v_graph_create(g.row_ptr.ctypes.data, g.col_idx.ctypes.data, None, None, 34,
                                   len(g.col_idx), 34, 0, 0, C.byref(handle)))
    out = torch.empty((68, 20), dtype=torch.int32, device="cuda")
    wp = _lib.WalkParams(20, 2, 1.0, 1.0, 100, 0, 5.0, 0.2)
    stream = torch.cuda.current_stream().cuda_stream

    def run():
        _lib.check(L.gn2v_walks(handle, C.byref(wp), 3, 0, 0, 68, out.data_ptr(), stream))
        return _u32(out).copy()

    untyped = run()
    _lib.check(L.gn2v_graph_set_types(handle, g.node_type_ids.ctypes.data, g.edge_type_ids.ctypes.data))
    typed = run()
    assert np.array_equal(typed, O.walks(oracle_graph(g), O.WalkParams(20, 2, 1.0, 1.0, 100, 0, 5.0, 0.2), 3, 0, 0, 68))
    assert not np.array_equal(typed, untyped)
    _lib.check(L.gn2v_graph_set_types(handle, None, None))  # detach
    assert np.array_equal(run(), untyped)
    bad = _lib.WalkParams(20, 2, 1.0, 1.0, 100, 0, -1.0, 1.0)
    assert L.gn2v_walks(handle, C.byref(bad), 3, 0, 0, 68, out.data_ptr(), stream) != 0
    assert b"change_node_type_weight" in L.gn2v_last_error()
    assert L.gn2v_graph_set_types(None, None, None) != 0
    L.gn2v_graph_destroy(handle)


@pytest.mark.parametrize("rw,ew,cn,ce", [(1.0, 1.0, 4.0, 1.0), (1.0, 1.0, 1.0, 0.2),
                                         (0.25, 4.0, 2.0, 0.5), (2.0, 0.5, 2.0, 0.5),
                                         (1e-3, 1e-3, 1e-3, 1e-3)])
def test_typed_edge_records_only_accelerate_the_reads(monkeypatch, rw, ew, cn, ce):
    """Walks with type factors read 32 B edge records (the 16 B record, then the node type of the
    destination and the type of the edge; walk_kernels.h walk_rec_kernel<true>): the same walks
    with the records off and in the oracle, on a multigraph whose parallel edges differ in type."""
    rng = np.random.RandomState(8)
    s, d = O.ba_edges(3000, 5, 4)
    extra = rng.randint(0, len(s), size=2000)
    src, dst = np.concatenate([s, s[extra]]), np.concatenate([d, d[extra]])
    et = np.concatenate([rng.randint(0, 3, size=len(s)), rng.randint(3, 5, size=len(extra))])
    nt = rng.randint(0, 4, size=3000)
    walks = {}
    for records in ("1", "0"):
        monkeypatch.setenv("GN2V_WALK_EDGE_RECORDS", records)
        g = E.CSRGraph.from_edge_list(src, dst, number_of_nodes=3000, node_types=nt.tolist(),
                                      edge_types=et.tolist())
        walks[records] = _u32(ops.walks(g, ops.walk_params(40, 2, rw, ew, 100, cn, ce), 3, 1, 7, 6100))
        assert ops.walk_accel(g) & 12 == (ops.WALK_ACCEL_TYPED_RECORDS if records == "1" else 0)
    assert np.array_equal(walks["1"], walks["0"])
    assert np.array_equal(walks["1"], O.walks(oracle_graph(g), O.WalkParams(40, 2, rw, ew, 100, 0, cn, ce),
                                              3, 1, 7, 6100))


def test_new_types_rebuild_the_typed_records():
    import ctypes as C

    g = typed_karate()
    L = _lib.lib()
    handle = C.c_void_p()
    _lib.check(L.gn2v_graph_create(g.row_ptr.ctypes.data, g.col_idx.ctypes.data, None, None, 34,
                                   len(g.col_idx), 34, 0, 0, C.byref(handle)))
    out = torch.empty((340, 20), dtype=torch.int32, device="cuda")
    wp = _lib.WalkParams(20, 2, 0.5, 2.0, 100, 0, 5.0, 0.2)
    owp = O.WalkParams(20, 2, 0.5, 2.0, 100, 0, 5.0, 0.2)
    stream = torch.cuda.current_stream().cuda_stream
    rng = np.random.RandomState(2)
    for trial in range(3):
        nt = rng.randint(0, 3, size=34).astype(np.uint32)
        et = rng.randint(0, 4, size=len(g.col_idx)).astype(np.uint32)
        _lib.check(L.gn2v_graph_set_types(handle, nt.ctypes.data, et.ctypes.data))
        assert L.gn2v_graph_walk_accel(handle) & ops.WALK_ACCEL_TYPED_RECORDS == 0
        _lib.check(L.gn2v_walks(handle, C.byref(wp), 3, trial, 0, 340, out.data_ptr(), stream))
        assert L.gn2v_graph_walk_accel(handle) & ops.WALK_ACCEL_TYPED_RECORDS
        og = O.OracleGraph(g.row_ptr, g.col_idx, None, nt, et)
        assert np.array_equal(_u32(out), O.walks(og, owp, 3, trial, 0, 340))
    L.gn2v_graph_destroy(handle)

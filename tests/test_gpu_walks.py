"""GPU parity: the HIP walk sampler is bit-identical to the oracle (integer work => exact)."""
import numpy as np
import pytest
import torch

import embiggen_amd as E
from embiggen_amd import _lib, ops
from oracle import oracle as O

pytestmark = pytest.mark.gpu


def _u32(t):
    return t.cpu().numpy().view(np.uint32)


@pytest.mark.parametrize("rw,ew", [(1.0, 1.0), (0.25, 4.0), (2.0, 0.5), (4.0, 0.25),
                                   (1e-3, 1e-3), (1.0, 1e-4)])
@pytest.mark.parametrize("walk_length", [2, 5, 16, 33, 128])
def test_walks_bit_exact_on_karate(karate, karate_oracle, rw, ew, walk_length):
    wp = ops.walk_params(walk_length, 10, rw, ew)
    got = _u32(ops.walks(karate, wp, 42, 3, 0, 340))
    ref = O.walks(karate_oracle, O.WalkParams(walk_length, 10, rw, ew, 100, 0), 42, 3, 0, 340)
    assert np.array_equal(got, ref)


def test_walks_batch_offsets_and_epochs(karate, karate_oracle):
    wp = ops.walk_params(20, 3, 0.5, 2.0)
    owp = O.WalkParams(20, 3, 0.5, 2.0, 100, 0)
    whole = _u32(ops.walks(karate, wp, 1, 2, 0, 102))
    parts = np.concatenate([_u32(ops.walks(karate, wp, 1, 2, 0, 37)),
                            _u32(ops.walks(karate, wp, 1, 2, 37, 65))])
    assert np.array_equal(whole, parts)
    assert np.array_equal(whole, O.walks(karate_oracle, owp, 1, 2, 0, 102))
    # ragged wave: 1 walk, 63, 65 walks
    for n in (1, 63, 65):
        assert np.array_equal(_u32(ops.walks(karate, wp, 9, 0, 5, n)),
                              O.walks(karate_oracle, owp, 9, 0, 5, n))


def test_walks_bit_exact_on_ba_graph():
    """Scale-free graph with hubs (binary search over long adjacency lists)."""
    s, d = O.ba_edges(5000, 4, 11)
    g = E.CSRGraph.from_edge_list(s, d, number_of_nodes=5000)
    og = O.OracleGraph(g.row_ptr, g.col_idx)
    for rw, ew in ((0.25, 4.0), (2.0, 0.5)):
        got = _u32(ops.walks(g, ops.walk_params(64, 2, rw, ew), 5, 1, 0, 10000))
        assert np.array_equal(got, O.walks(og, O.WalkParams(64, 2, rw, ew, 100, 0), 5, 1, 0, 10000))


@pytest.mark.parametrize("rw,ew", [(0.25, 4.0), (2.0, 0.5), (4.0, 0.25)])
def test_edge_set_and_filter_only_accelerate_the_adjacency_test(monkeypatch, rw, ew):
    """The second-order sampler asks "is x adjacent to prev" of a hash set of the edges behind a
    three-bit filter (built on a handle's first biased walk) instead of binary searches: the
    same walks with the filter switched off, with the set switched off (binary searches, as in
    rounds 1-3), and in the oracle."""
    s, d = O.ba_edges(5000, 4, 11)
    owp = O.WalkParams(40, 2, rw, ew, 100, 0)
    ref = None
    for env, accel in (({}, 7), ({"GN2V_WALK_EDGE_FILTER": "0"}, 5),
                       ({"GN2V_WALK_EDGE_SET": "0"}, 4), ({"GN2V_WALK_EDGE_RECORDS": "0"}, 3),
                       ({"GN2V_WALK_EDGE_RECORDS": "0", "GN2V_WALK_EDGE_SET": "0"}, 0)):
        for k in ("GN2V_WALK_EDGE_FILTER", "GN2V_WALK_EDGE_SET", "GN2V_WALK_EDGE_RECORDS"):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        g = E.CSRGraph.from_edge_list(s, d, number_of_nodes=5000)  # a handle of its own
        got = _u32(ops.walks(g, ops.walk_params(40, 2, rw, ew), 7, 1, 0, 10_000))
        assert ops.walk_accel(g) == accel, env
        if ref is None:
            ref = O.walks(O.OracleGraph(g.row_ptr, g.col_idx), owp, 7, 1, 0, 10_000)
        assert np.array_equal(got, ref), env


@pytest.mark.parametrize("directed", [False, True])
@pytest.mark.parametrize("rw,ew", [(1.0, 1.0), (0.25, 4.0), (2.0, 0.5), (4.0, 0.25), (1e-3, 1e-3)])
def test_edge_records_only_accelerate_the_reads(monkeypatch, rw, ew, directed):
    """An unweighted walk without type factors reads one 16 B edge record per candidate
    (destination, its row, the signature of its neighbourhood) instead of row_ptr + col_idx + the
    adjacency probe (walk_kernels.h walk_rec_kernel): the same walks with the records switched
    off and in the oracle -- on a symmetric graph (prev is a neighbour for certain when it is
    proposed on its own) and on a directed one with traps (it is looked up), hubs whose signature
    is full, first and second order, both envelopes."""
    rng = np.random.RandomState(5)
    if directed:
        src, dst = rng.randint(0, 2000, size=12_000), rng.randint(0, 2000, size=12_000)
        make = lambda: E.CSRGraph.from_edge_list(src, dst, number_of_nodes=2100, directed=True)
    else:
        s, d = O.ba_edges(4000, 6, 3)
        make = lambda: E.CSRGraph.from_edge_list(s, d, number_of_nodes=4000)
    owp = O.WalkParams(48, 2, rw, ew, 100, 0)
    walks = {}
    for records in ("1", "0"):
        monkeypatch.setenv("GN2V_WALK_EDGE_RECORDS", records)
        g = make()
        n = 2 * g.get_number_of_unique_source_nodes() + 37
        walks[records] = _u32(ops.walks(g, ops.walk_params(48, 2, rw, ew), 11, 2, 5, n))
        assert bool(ops.walk_accel(g) & ops.WALK_ACCEL_RECORDS) == (records == "1")
    assert np.array_equal(walks["1"], walks["0"])
    og = O.OracleGraph(g.row_ptr, g.col_idx)
    assert np.array_equal(walks["1"], O.walks(og, owp, 11, 2, 5, n, sources=g.sources))


@pytest.mark.parametrize("rw,ew", [(1.0, 1.0), (0.25, 4.0), (2.0, 0.5)])
def test_edge_records_on_a_weighted_graph(monkeypatch, rw, ew):
    """Weighted graphs find the candidate in the row's cumulative weights and read its edge record
    (the row of the node moved to, the signature for the adjacency test): the same walks with the
    records off and in the oracle."""
    rng = np.random.RandomState(4)
    s, d = O.ba_edges(3000, 5, 6)
    w = rng.uniform(0.05, 6.0, size=len(s))
    walks = {}
    for records in ("1", "0"):
        monkeypatch.setenv("GN2V_WALK_EDGE_RECORDS", records)
        g = E.CSRGraph.from_edge_list(s, d, w, number_of_nodes=3000)
        walks[records] = _u32(ops.walks(g, ops.walk_params(40, 2, rw, ew), 9, 1, 3, 6100))
        assert bool(ops.walk_accel(g) & ops.WALK_ACCEL_RECORDS) == (records == "1")
    assert np.array_equal(walks["1"], walks["0"])
    og = O.OracleGraph(g.row_ptr, g.col_idx, g.cumw)
    assert np.array_equal(walks["1"], O.walks(og, O.WalkParams(40, 2, rw, ew, 100, 0), 9, 1, 3, 6100))


def test_weighted_walks_bit_exact():
    rng = np.random.RandomState(3)
    s, d = O.ba_edges(800, 3, 5)
    w = rng.uniform(0.1, 5.0, size=len(s))
    g = E.CSRGraph.from_edge_list(s, d, w, number_of_nodes=800)
    og = O.OracleGraph(g.row_ptr, g.col_idx, g.cumw)
    for rw, ew in ((1.0, 1.0), (0.25, 4.0), (1.0, 1e-4)):
        got = _u32(ops.walks(g, ops.walk_params(24, 2, rw, ew), 8, 0, 0, 1600))
        assert np.array_equal(got, O.walks(og, O.WalkParams(24, 2, rw, ew, 100, 0), 8, 0, 0, 1600))


def test_directed_graph_with_traps_and_isolated_nodes():
    rng = np.random.RandomState(0)
    src = rng.randint(0, 300, size=900)
    dst = rng.randint(0, 300, size=900)
    g = E.CSRGraph.from_edge_list(src, dst, number_of_nodes=320, directed=True)
    assert g.sources is not None and g.has_disconnected_nodes()
    og = O.OracleGraph(g.row_ptr, g.col_idx)
    n = 2 * g.get_number_of_unique_source_nodes()
    got = _u32(ops.walks(g, ops.walk_params(40, 2, 0.5, 2.0), 4, 0, 0, n))
    ref = O.walks(og, O.WalkParams(40, 2, 0.5, 2.0, 100, 0), 4, 0, 0, n, sources=g.sources)
    assert np.array_equal(got, ref)
    assert (got == _lib.SENTINEL).any()  # some walks did hit a trap


def test_ba_edges_and_init_table_bit_exact():
    n, m = 3000, 7
    src = torch.empty((n - 1) * m, dtype=torch.int32, device="cuda")
    dst = torch.empty_like(src)
    _lib.check(_lib.lib().gn2v_ba_edges(n, m, 42, src.data_ptr(), dst.data_ptr(), None))
    torch.cuda.synchronize()
    rs, rd = O.ba_edges(n, m, 42)
    assert np.array_equal(_u32(src), rs) and np.array_equal(_u32(dst), rd)
    for d, tid, scale in ((5, 0, 0.4), (128, 1, 128 ** -0.5), (100, 1, 0.1)):
        ld = (d + 3) // 4 * 4
        got = ops.init_table(77, d, 9, tid, scale).cpu().numpy()
        assert np.array_equal(got, O.init_table(77, d, ld, 9, tid, scale))
        assert (got[:, d:] == 0).all()


def test_device_built_ba_graph_equals_host_build():
    g = E.barabasi_albert(4000, 5, 42)
    s, d = O.ba_edges(4000, 5, 42)
    h = E.CSRGraph.from_edge_list(s, d, number_of_nodes=4000)
    assert np.array_equal(g.row_ptr, h.row_ptr) and np.array_equal(g.col_idx, h.col_idx)
    assert g.get_number_of_nodes() == 4000 and g.sources is None


def test_window_batch_matches_oracle(karate, karate_oracle):
    """Node2VecSequence batch form (node2vec_sequence.py:115-128,190-203)."""
    wk = ops.walks(karate, ops.walk_params(128, 16, 1.0, 1.0), 42, 0, 0, 34 * 16)
    contexts, words = ops.window_batch(wk, 4)
    rc, rw = O.window_batch(_u32(wk), 4)
    assert contexts.shape == (34 * 16 * 120, 8) and words.shape == (34 * 16 * 120,)
    assert np.array_equal(contexts.cpu().numpy(), rc) and np.array_equal(words.cpu().numpy(), rw)


def test_full_size_walk_properties():
    """At benchmark scale (no oracle run): every step follows an edge, starts are the sources,
    and a re-run is identical."""
    g = E.barabasi_albert(1_000_000, 10, 42)
    wp = ops.walk_params(128, 10, 0.25, 4.0)
    wk = ops.walks(g, wp, 42, 0, 0, 1 << 15)
    again = ops.walks(g, wp, 42, 0, 0, 1 << 15)
    assert torch.equal(wk, again)
    t = g._device_tensors
    a, b = wk[:, :-1].long().flatten(), wk[:, 1:].long().flatten()
    n = g.get_number_of_nodes()
    edge_keys = (torch.arange(n, device="cuda").repeat_interleave(
        t["row_ptr"][1:] - t["row_ptr"][:-1]) * n + t["col_idx"].long())
    pos = torch.searchsorted(edge_keys, a * n + b).clamp_(max=edge_keys.numel() - 1)
    assert bool((edge_keys[pos] == a * n + b).all())
    assert torch.equal(wk[:, 0].long(), torch.arange(1 << 15, device="cuda") % n)


def test_normalize_by_degree_walks_and_fit(karate):
    """normalize_by_degree through the public API: host graph and device-resident graph, walks
    bit-identical to the oracle fed with the same derived weights."""
    h = karate.with_degree_normalized_weights()
    og = O.OracleGraph(h.row_ptr, h.col_idx, h.cumw)
    for rw, ew in ((1.0, 1.0), (0.25, 4.0)):
        got = _u32(ops.walks(h, ops.walk_params(20, 3, rw, ew), 6, 0, 0, 102))
        assert np.array_equal(got, O.walks(og, O.WalkParams(20, 3, rw, ew, 100, 0), 6, 0, 0, 102))
    g = E.barabasi_albert(3000, 4, 9)
    hd = g.with_degree_normalized_weights()
    assert hd.has_edge_weights() and hd is g.with_degree_normalized_weights()
    ogd = O.OracleGraph(hd.row_ptr, hd.col_idx, hd.cumw)
    got = _u32(ops.walks(hd, ops.walk_params(16, 1, 0.5, 2.0), 6, 0, 0, 3000))
    assert np.array_equal(got, O.walks(ogd, O.WalkParams(16, 1, 0.5, 2.0, 100, 0), 6, 0, 0, 3000))
    m = E.Node2VecSkipGramEnsmallen(embedding_size=8, epochs=1, walk_length=8, iterations=1,
                                    normalize_by_degree=True, verbose=False)
    res = m.fit_transform(karate, return_dataframe=False).get_all_node_embedding()
    plain = E.Node2VecSkipGramEnsmallen(embedding_size=8, epochs=1, walk_length=8, iterations=1,
                                        verbose=False).fit_transform(karate, return_dataframe=False)
    assert np.isfinite(res[0]).all()
    assert not np.array_equal(res[0], plain.get_all_node_embedding()[0])


def test_max_neighbours_of_the_smoke_configuration_reaches_the_sampler(karate, karate_oracle):
    """The smoke configuration of the reference sets ``max_neighbours=10``
    (embedders/ensmallen_embedders/node2vec.py:79-87; meaning: node2vec_skipgram.py:78-81).
    Karate's hubs have 12-17 neighbours: with 10 (and 3) their steps are taken over a sub-sample,
    with 100 and None the walks are exact -- on the GPU, through the C ABI, as in the oracle."""
    import embiggen_amd as E

    smoke = E.Node2VecSkipGramEnsmallen().into_smoke_test()
    assert smoke.parameters()["max_neighbours"] == 10
    res = smoke.fit_transform(karate, return_dataframe=False).get_all_node_embedding()
    assert res[0].shape == (34, 5) and np.isfinite(res[0]).all()
    exact = O.walks(karate_oracle, O.WalkParams(16, 2, 0.25, 4.0, 0, 0), 9, 0, 0, 68)
    for mn in (10, 100, None, 3):
        wk = ops.walks(karate, ops.walk_params(16, 2, 0.25, 4.0, max_neighbours=mn), 9, 0, 0, 68)
        ref = O.walks(karate_oracle, O.WalkParams(16, 2, 0.25, 4.0, mn or 0, 0), 9, 0, 0, 68)
        assert np.array_equal(wk.cpu().numpy().view(np.uint32), ref)
        assert np.array_equal(ref, exact) == (mn in (100, None))


# ---------------------------------------------------------------------- max_neighbours
@pytest.mark.parametrize("records", ["1", "0"])
@pytest.mark.parametrize("max_neighbours", [3, 10, 100, None])
@pytest.mark.parametrize("rw,ew", [(1.0, 1.0), (0.25, 4.0), (2.0, 0.5), (4.0, 0.25)])
def test_max_neighbours_sub_sampled_walks_bit_exact(monkeypatch, rw, ew, max_neighbours, records):
    """``max_neighbours`` (node2vec_skipgram.py:22,78-81): steps out of nodes of higher degree
    choose among a per-visit sub-sample of that many edges (walk_kernels.h RowView; the oracle's
    row_view, whose law tests/test_oracle.py checks against an enumeration of every sub-sample).
    Bit-exact against the oracle for 3, 10, 100 (the default) and None (exact walks), on a
    scale-free graph whose hubs exceed 100 neighbours and on a directed graph with traps, through
    the CSR kernel and the edge records, first and second order, both envelopes of the rejection
    sampler (return apart: the previous node counts only when its bucket drew it)."""
    monkeypatch.setenv("GN2V_WALK_EDGE_RECORDS", records)
    m = 0 if max_neighbours is None else max_neighbours
    s, d = O.ba_edges(4000, 6, 3)
    g = E.CSRGraph.from_edge_list(s, d, number_of_nodes=4000)
    deg = np.diff(g.row_ptr.astype(np.int64))
    assert deg.max() > 100 and (deg > 10).sum() > 200
    og = O.OracleGraph(g.row_ptr, g.col_idx)
    got = _u32(ops.walks(g, ops.walk_params(48, 2, rw, ew, max_neighbours), 11, 2, 5, 8100))
    ref = O.walks(og, O.WalkParams(48, 2, rw, ew, m, 0), 11, 2, 5, 8100)
    assert np.array_equal(got, ref)
    if max_neighbours is not None:  # the sub-sample acts: other walks than the exact ones
        assert not np.array_equal(ref, O.walks(og, O.WalkParams(48, 2, rw, ew, 0, 0), 11, 2, 5, 8100))
    rng = np.random.RandomState(5)
    src, dst = rng.randint(0, 500, size=12_000), rng.randint(0, 500, size=12_000)
    gd = E.CSRGraph.from_edge_list(src, dst, number_of_nodes=520, directed=True)
    n = 2 * gd.get_number_of_unique_source_nodes() + 37
    got = _u32(ops.walks(gd, ops.walk_params(40, 2, rw, ew, max_neighbours), 3, 1, 0, n))
    assert np.array_equal(got, O.walks(O.OracleGraph(gd.row_ptr, gd.col_idx),
                                       O.WalkParams(40, 2, rw, ew, m, 0), 3, 1, 0, n,
                                       sources=gd.sources))


@pytest.mark.parametrize("records", ["1", "0"])
@pytest.mark.parametrize("max_neighbours", [4, 100])
def test_max_neighbours_on_weighted_and_typed_graphs(monkeypatch, max_neighbours, records):
    """Weights on a sub-sample have no cumulative sums to search: the step is the exact scan over
    the sub-sample's elements (weight x threshold) -- unbiased steps included; typed graphs draw
    their candidates from the sub-sample and apply the type factors as before."""
    monkeypatch.setenv("GN2V_WALK_EDGE_RECORDS", records)
    rng = np.random.RandomState(4)
    s, d = O.ba_edges(3000, 5, 6)
    w = rng.uniform(0.05, 6.0, size=len(s))
    g = E.CSRGraph.from_edge_list(s, d, w, number_of_nodes=3000)
    og = O.OracleGraph(g.row_ptr, g.col_idx, g.cumw)
    for rw, ew in ((1.0, 1.0), (0.25, 4.0), (2.0, 0.5)):
        got = _u32(ops.walks(g, ops.walk_params(32, 2, rw, ew, max_neighbours), 9, 1, 3, 6100))
        assert np.array_equal(got, O.walks(og, O.WalkParams(32, 2, rw, ew, max_neighbours, 0),
                                           9, 1, 3, 6100)), (rw, ew)
    gt = E.CSRGraph.from_edge_list(s, d, number_of_nodes=3000,
                                   node_types=rng.randint(0, 4, size=3000).tolist(),
                                   edge_types=rng.randint(0, 3, size=len(s)).tolist())
    ogt = O.OracleGraph(gt.row_ptr, gt.col_idx, gt.cumw, gt.node_type_ids, gt.edge_type_ids)
    for rw, ew, cn, ce in ((0.25, 4.0, 2.0, 0.5), (1.0, 1.0, 0.1, 10.0), (2.0, 0.5, 1.0, 3.0)):
        got = _u32(ops.walks(gt, ops.walk_params(32, 2, rw, ew, max_neighbours, cn, ce), 5, 1, 0,
                             6000))
        assert np.array_equal(got, O.walks(ogt, O.WalkParams(32, 2, rw, ew, max_neighbours, 0, cn,
                                                             ce), 5, 1, 0, 6000)), (rw, ew, cn, ce)
    gw = E.CSRGraph.from_edge_list(s, d, w, number_of_nodes=3000,
                                   node_types=rng.randint(0, 3, size=3000).tolist(),
                                   edge_types=rng.randint(0, 3, size=len(s)).tolist())
    ogw = O.OracleGraph(gw.row_ptr, gw.col_idx, gw.cumw, gw.node_type_ids, gw.edge_type_ids)
    got = _u32(ops.walks(gw, ops.walk_params(32, 2, 0.25, 4.0, max_neighbours, 0.5, 2.0), 5, 1, 0,
                         6000))
    assert np.array_equal(got, O.walks(ogw, O.WalkParams(32, 2, 0.25, 4.0, max_neighbours, 0, 0.5,
                                                         2.0), 5, 1, 0, 6000))

"""BASELINE.json configs 4 and 5 at their full sizes on one MI355X, through size-independent
properties (the oracle cannot run these sizes in seconds):

  * walks follow edges of the CSR (u64 ``row_ptr``: config 5 has 2 x 10^9 directed edges) and
    re-run identically;
  * the pair / centre counters of a training launch equal the closed form 2wL - w(w+1);
  * a zero learning rate is the identity (bit for bit);
  * central rows move only for nodes on the walks, contextual rows also for the negatives;
  * everything stays finite.

Tables are checked through exact per-row integer checksums so that the 2 x 51.2 GB tables of
config 5 need no second copy.  Config 4 additionally runs the block-partitioned multi-GPU trainer
with 8 simulated ranks at full size (tests/test_gpu_blocks.py holds the small exact cases).

``block_path_properties`` drives the kernel that SHIPS at these sizes -- SkipGram on >= 2^16 nodes
goes through ``gn2v_train`` -> ``gn2v_train_blocks`` (automatic plan, alias tables, pair words,
groups of parts, ``sgns_block_kernel``, the contextual table trained part-major in the caller's
buffer and put back in node order) -- for configs 3, 4, 5a and 5 with a walk budget;
``full_size_properties`` keeps the walk-ordered kernel (CBOW's, small graphs', explicit update
modes') honest at the same sizes.
"""
import numpy as np
import pytest
import torch

import embiggen_amd as E
from embiggen_amd import ops

pytestmark = pytest.mark.gpu

PAIRS_PER_WALK = 2 * 5 * 128 - 5 * 6  # window 5, walk_length 128


def row_checksums(t, chunk=1 << 21):
    """Exact int64 checksum of every row (bit pattern sums), computed in row chunks."""
    out = torch.empty(t.shape[0], dtype=torch.int64, device=t.device)
    for lo in range(0, t.shape[0], chunk):
        out[lo:lo + chunk] = t[lo:lo + chunk].view(torch.int32).sum(1, dtype=torch.int64)
    return out


def all_finite(t, chunk=1 << 23):
    return all(bool(torch.isfinite(t[lo:lo + chunk]).all()) for lo in range(0, t.shape[0], chunk))


def walks_follow_edges(g, wk, chunk=1 << 22):
    """Every consecutive walk pair (a, b) has b in the sorted adjacency row of a."""
    t = g._device_tensors
    col, row_ptr = t["col_idx"], t["row_ptr"]
    a_all, b_all = wk[:, :-1].reshape(-1), wk[:, 1:].reshape(-1)
    for off in range(0, a_all.numel(), chunk):
        a = a_all[off:off + chunk].long() & 0xFFFFFFFF
        b = b_all[off:off + chunk].long() & 0xFFFFFFFF
        lo, ends = row_ptr[a].clone(), row_ptr[a + 1]
        hi = ends.clone()
        for _ in range(32):
            mid = (lo + hi) // 2
            val = col[mid.clamp(max=col.numel() - 1)].long() & 0xFFFFFFFF
            go_right = (mid < hi) & (val < b)
            lo = torch.where(go_right, mid + 1, lo)
            hi = torch.where(go_right, hi, mid)
        hit = col[lo.clamp(max=col.numel() - 1)].long() & 0xFFFFFFFF
        if not bool(((lo < ends) & (hit == b)).all()):
            return False
    return True


def full_size_properties(g, n_walks, d=128):
    n = g.get_number_of_nodes()
    wp = ops.walk_params(128, 10, 0.25, 4.0)
    wk = ops.walks(g, wp, 42, 0, 0, n_walks)
    assert torch.equal(wk, ops.walks(g, wp, 42, 0, 0, n_walks))
    assert int((wk == -1).sum()) == 0  # BA graphs have no trap nodes
    assert walks_follow_edges(g, wk)

    c = ops.init_table(n, d, 42, 0, d ** -0.5)
    x = ops.init_table(n, d, 42, 1, d ** -0.5)
    c0, x0 = row_checksums(c), row_checksums(x)
    tp = ops.train_params(0, d, 10, 5)
    ops.sgns_step(g, tp, wk, 42, 0, 0, 0.0, c, x)  # lr = 0: identity
    assert torch.equal(row_checksums(c), c0) and torch.equal(row_checksums(x), x0)
    ops.stats_reset(g)
    ops.sgns_step(g, tp, wk, 42, 0, 0, 0.01, c, x)
    st = ops.stats_read(g)
    assert st["pairs"] == n_walks * PAIRS_PER_WALK and st["centres"] == n_walks * 128
    assert all_finite(c) and all_finite(x)
    visited = torch.zeros(n, dtype=torch.bool, device="cuda")
    visited[wk.long().flatten() & 0xFFFFFFFF] = True
    c1, x1 = row_checksums(c), row_checksums(x)
    assert torch.equal(c1[~visited], c0[~visited])  # central rows move only for walk nodes
    assert float((c1[visited] != c0[visited]).float().mean()) > 0.99
    moved = int((x1 != x0).sum())
    assert moved > int(visited.sum())  # negatives reach beyond the walk nodes
    return st


def block_path_properties(g, n_walks, plan, d=128, return_weight=0.25, explore_weight=4.0):
    """The whole fit through ``gn2v_train`` (block path by the library's own choice) on a budget of
    ``n_walks`` walks: lr = 0 is the identity (row checksums against the freshly initialised
    tables, in NODE order: the part-major storage of the fit is undone); pairs == the closed
    form; central rows move only for walk nodes and nearly every walk node's row moves; the
    contextual rows of (nearly) all walk nodes move and the negatives reach beyond them; finite."""
    n = g.get_number_of_nodes()
    kw = dict(embedding_size=d, epochs=1, walk_length=128, iterations=10, window_size=5,
              number_of_negative_samples=10, return_weight=return_weight,
              explore_weight=explore_weight, verbose=False)
    scale = d ** -0.5
    c0 = row_checksums(ops.init_table(n, d, 42, 0, scale))
    x0 = row_checksums(ops.init_table(n, d, 42, 1, scale))
    torch.cuda.empty_cache()
    m = E.models.SkipGram(learning_rate=0.0, **kw)
    c, x, st = m.fit_transform_device(g, max_walks_per_epoch=n_walks)
    got = {k: m.last_plan[k] for k in ("parts", "slices")}
    assert got == plan, m.last_plan
    assert st["pairs"] == n_walks * PAIRS_PER_WALK
    assert torch.equal(row_checksums(c), c0) and torch.equal(row_checksums(x), x0)
    del c, x
    torch.cuda.empty_cache()
    m = E.models.SkipGram(learning_rate=0.01, **kw)
    c, x, st = m.fit_transform_device(g, max_walks_per_epoch=n_walks)
    assert st["pairs"] == n_walks * PAIRS_PER_WALK
    assert 0 < st["centres"] <= st["pairs"]  # runs of equal centre inside the records
    assert all_finite(c) and all_finite(x)
    wk = ops.walks(g, m.walk_params(), 42, 0, 0, n_walks)  # the walks of that fit
    visited = torch.zeros(n, dtype=torch.bool, device="cuda")
    visited[wk.long().flatten() & 0xFFFFFFFF] = True
    c1, x1 = row_checksums(c), row_checksums(x)
    assert torch.equal(c1[~visited], c0[~visited])  # central rows move only for walk nodes
    assert float((c1[visited] != c0[visited]).float().mean()) > 0.99
    assert float((x1[visited] != x0[visited]).float().mean()) > 0.99  # every walk node is a context
    assert int((x1 != x0).sum()) > int(visited.sum())  # negatives reach beyond the walk nodes
    return m.last_plan, st


def cbow_full_size_properties(g, n_walks, d=128):
    """CBOW (embedders/ensmallen_embedders/node2vec_cbow.py:9-146) at full size: the mirror image
    of SkipGram's properties -- the mean of a window's CONTEXTUAL rows is scored against the
    central rows of the centre and of k negatives, every context row receives the shared
    gradient: contextual rows move only for walk nodes, central rows also for the negatives.
    First one launch (``gn2v_cbow_step``), then the whole fit through ``gn2v_train`` -- the call
    ``Node2VecCBOWEnsmallen.fit_transform`` makes -- on a budget of ``n_walks`` walks."""
    n = g.get_number_of_nodes()
    wp = ops.walk_params(128, 10, 0.25, 4.0)
    wk = ops.walks(g, wp, 42, 0, 0, n_walks)
    c = ops.init_table(n, d, 42, 0, d ** -0.5)
    x = ops.init_table(n, d, 42, 1, d ** -0.5)
    c0, x0 = row_checksums(c), row_checksums(x)
    tp = ops.train_params(1, d, 10, 5)
    ops.cbow_step(g, tp, wk, 42, 0, 0, 0.0, c, x)  # lr = 0: identity
    assert torch.equal(row_checksums(c), c0) and torch.equal(row_checksums(x), x0)
    ops.stats_reset(g)
    ops.cbow_step(g, tp, wk, 42, 0, 0, 0.01, c, x)
    st = ops.stats_read(g)
    # every walk position is a centre; its contexts are the pairs of the closed form
    assert st["centres"] == n_walks * 128 and st["pairs"] == n_walks * PAIRS_PER_WALK, st
    assert all_finite(c) and all_finite(x)
    visited = torch.zeros(n, dtype=torch.bool, device="cuda")
    visited[wk.long().flatten() & 0xFFFFFFFF] = True
    c1, x1 = row_checksums(c), row_checksums(x)
    assert torch.equal(x1[~visited], x0[~visited])  # contextual rows move only for walk nodes
    assert float((x1[visited] != x0[visited]).float().mean()) > 0.99
    assert float((c1[visited] != c0[visited]).float().mean()) > 0.99  # every walk node is a centre
    assert int((c1 != c0).sum()) > int(visited.sum())  # negatives reach beyond the walk nodes
    del c, x, c1, x1
    torch.cuda.empty_cache()
    kw = dict(embedding_size=d, epochs=1, walk_length=128, iterations=10, window_size=5,
              number_of_negative_samples=10, verbose=False)
    m = E.models.CBOW(learning_rate=0.0, **kw)
    c, x, st = m.fit_transform_device(g, max_walks_per_epoch=n_walks)
    assert m.last_plan is None  # CBOW has no block path
    assert st["centres"] == n_walks * 128 and st["pairs"] == n_walks * PAIRS_PER_WALK, st
    assert torch.equal(row_checksums(c), c0) and torch.equal(row_checksums(x), x0)
    del c, x
    torch.cuda.empty_cache()
    m = E.models.CBOW(learning_rate=0.01, **kw)
    c, x, st = m.fit_transform_device(g, max_walks_per_epoch=n_walks)
    assert st["centres"] == n_walks * 128 and all_finite(c) and all_finite(x)
    x1 = row_checksums(x)
    assert torch.equal(x1[~visited], x0[~visited])
    assert float((x1[visited] != x0[visited]).float().mean()) > 0.99
    return st


def test_config4_products_shaped_cbow_full_size_properties():
    """BASELINE config 4's shape (2 449 029 nodes / 61 M edges), CBOW, d = 128."""
    g = E.barabasi_albert(2_449_029, 25, 42, name="BA-shaped-like-ogbn-products")
    cbow_full_size_properties(g, 1 << 16)


def test_config5a_bench_graph_cbow_full_size_properties():
    """The roofline configuration (BA 10 M / 100 M), CBOW, d = 128: what ``bench.py --model cbow``
    times."""
    g = E.barabasi_albert(10_000_000, 10, 42)
    cbow_full_size_properties(g, 1 << 16)


def test_config3_arxiv_shaped_block_path_full_size_properties():
    """BASELINE config 3's shape (169 343 nodes, p = 0.5 / q = 2): resident cells since round 4
    -- 4 parts x 256 cells of 166 rows, each trained by one workgroup in its LDS (rounds 2-3: 1 x 8
    XCD cells, racing stores)."""
    g = E.barabasi_albert(169_343, 7, 42, name="BA-shaped-like-ogbn-arxiv")
    block_path_properties(g, 1 << 15, {"parts": 4, "slices": 256}, return_weight=2.0,
                          explore_weight=0.5)


def test_config4_products_shaped_block_path_full_size_properties():
    """BASELINE config 4's shape on one GPU: resident cells, 44 parts x 256 cells of 218 rows
    (up to round 4's cell limit of 8 192: 9 x 8 XCD cells)."""
    g = E.barabasi_albert(2_449_029, 25, 42, name="BA-shaped-like-ogbn-products")
    block_path_properties(g, 1 << 16, {"parts": 44, "slices": 256})


def test_config4_block_path_with_the_parts_trained_in_node_order(monkeypatch):
    """``GN2V_BLOCK_LAYOUT=natural``: the contextual table stays in node order during the fit,
    part p = its rows p, p + parts, ... (``gn2v_block_io.context_ld``) -- what ``gn2v_train_blocks``
    falls back to when the scratch copy of the part-major layout would not fit.  Same properties."""
    monkeypatch.setenv("GN2V_BLOCK_LAYOUT", "natural")
    g = E.barabasi_albert(2_449_029, 25, 42, name="BA-shaped-like-ogbn-products")
    block_path_properties(g, 1 << 16, {"parts": 44, "slices": 256})


def test_config5a_bench_graph_block_path_full_size_properties():
    """The roofline configuration (BA 10 M / 100 M) through the path the bench times: resident
    cells, 178 parts x 256 cells of 220 rows, a round's pairs extracted and trained in one group
    (one launch of all 45 568 cells)."""
    g = E.barabasi_albert(10_000_000, 10, 42)
    plan, _ = block_path_properties(g, 1 << 17, {"parts": 178, "slices": 256})
    assert plan["group_parts"] == 178


def test_config5a_bench_graph_with_xcd_cells(monkeypatch):
    """The same graph with the resident cells switched off (GN2V_RESIDENT_MAX_NODES): the XCD
    cells of rounds 2-4, 38 x 8 cells of 32.9 k rows -- what graphs beyond 13 M nodes, rows wider
    than 128 floats and graphs with a dominating hub keep running."""
    monkeypatch.setenv("GN2V_RESIDENT_MAX_NODES", "1500000")
    g = E.barabasi_albert(10_000_000, 10, 42)
    plan, _ = block_path_properties(g, 1 << 17, {"parts": 38, "slices": 8})
    assert plan["group_parts"] == 10  # four groups of parts per round


def test_config5_ba_100m_block_path_full_size_properties():
    """BASELINE config 5 (BA 100 M / 1 B) on one GPU: resident cells, 1 776 parts x 256 cells of
    220 rows (454 656 cells: 55-bit pair words, 19 + 27 + 9), extracted in wide groups of some 200
    parts (more cells than the counting pass has LDS counters: their offsets follow the sort),
    100 M-row alias tables rebuilt with every round's placement, the 51.2 GB contextual table
    trained where it lies, in node order."""
    g = E.barabasi_albert(100_000_000, 10, 42)
    plan, _ = block_path_properties(g, 1 << 17, {"parts": 1776, "slices": 256})
    assert 54 < plan["group_parts"] <= 256


def test_config5_ba_100m_with_xcd_cells(monkeypatch):
    """The same graph with the resident cells switched off: 381 x 8 XCD cells of 32.8 k rows,
    56-bit pair words (12 + 27 + 17) -- the plan of graphs beyond 115 M nodes."""
    monkeypatch.setenv("GN2V_RESIDENT_MAX_NODES", "1500000")
    g = E.barabasi_albert(100_000_000, 10, 42)
    plan, _ = block_path_properties(g, 1 << 17, {"parts": 381, "slices": 8})
    assert plan["group_parts"] == 96


def test_config4_ogbn_products_shaped_full_size_properties():
    """BASELINE config 4: ogbn-products-shaped BA graph, 2 449 029 nodes / ~61.2 M edges
    (m = 25), d = 128, reference defaults; one launch of 2^16 walks on one GPU."""
    g = E.barabasi_albert(2_449_029, 25, 42, name="BA-shaped-like-ogbn-products")
    assert g.get_number_of_nodes() == 2_449_029
    assert 2 * 60_000_000 < g.get_number_of_directed_edges() <= 2 * 61_225_700
    full_size_properties(g, 1 << 16)


def test_config5_ba_100m_nodes_1b_edges_full_size_properties():
    """BASELINE config 5: Barabasi-Albert 100 M nodes / 1 B edges (2 x 10^9 directed: u64 row
    pointers), d = 128: 2 x 51.2 GB tables + 8.8 GB CSR on one MI355X, one launch of 2^16 walks."""
    g = E.barabasi_albert(100_000_000, 10, 42)
    assert g.get_number_of_nodes() == 100_000_000
    e = g.get_number_of_directed_edges()
    assert 1.99e9 < e <= 2 * 999_999_990  # multi-edges collapse: a little under 2 x 10^9
    assert int(g._device_tensors["row_ptr"][-1]) == e
    full_size_properties(g, 1 << 16)


def test_config4_block_trainer_with_eight_simulated_ranks_at_full_size():
    """BASELINE config 4 ("embedding table row-sharded across 8 x MI355X") at its full size with
    the 8 ranks simulated on one GPU (exact: ranks never share a row): every pair of every round
    is trained exactly once, the context parts are each held by exactly one rank at the end,
    tables stay finite and the link quality equals the single trainer's on the same walks."""
    from embiggen_amd.distributed import BlockPartitionedTrainer, auto_plan
    from sharded_helpers import link_auc_device, run_ranks

    g = E.barabasi_albert(2_449_029, 25, 42, name="BA-shaped-like-ogbn-products")
    n, d, w, world = g.get_number_of_nodes(), 128, 5, 8
    wp = ops.walk_params(128, 10, 0.25, 4.0)
    total, per_round = 1 << 21, 1 << 15
    tp = ops.train_params(0, d, 10, w, flags=1, ld=d)
    c = ops.init_table(n, d, 42, 0, d ** -0.5)
    x = ops.init_table(n, d, 42, 1, d ** -0.5)
    for first in range(0, total, 1 << 16):
        ops.sgns_step(g, tp, ops.walks(g, wp, 42, 0, first, 1 << 16), 42, 0, first, 0.025, c, x)
    gen = torch.Generator(device="cuda")
    gen.manual_seed(1)
    auc_single = link_auc_device(g, c, x, gen)
    del c, x
    # resident cells travel too: 16 parts (two per rank) x 696 cells of <= 220 rows
    assert auto_plan(n, world, d, 10) == (16, 696) and auto_plan(n, world) == (16, 8)

    def rank_fn(comm):
        tr = BlockPartitionedTrainer(g, tp, d, d, 42, d ** -0.5, comm, "cuda:0", walk_length=128,
                                     window=w)
        trained = 0
        for r in range(total // per_round // world):
            first = r * world * per_round
            mine = ops.walks(g, wp, 42, 0, first + comm.rank * per_round, per_round)
            tr.train_round(mine, 42, 0, 0.025, first)
            trained += tr.last_round["pairs_trained"]
        held = sorted(tr.held)
        finite = all_finite(tr.central) and all(all_finite(t) for t in tr.held.values())
        full = tr.gather_full() if comm.rank == 0 else (tr.gather_full(), None)[1]
        return trained, held, finite, full

    ops.stats_reset(g)
    res = run_ranks(world, rank_fn)
    torch.cuda.synchronize()
    assert sum(r[0] for r in res) == total * PAIRS_PER_WALK == ops.stats_read(g)["pairs"]
    assert sorted(p for r in res for p in r[1]) == list(range(16))
    assert all(r[2] for r in res)
    bc, bx = res[0][3]
    gen.manual_seed(1)
    auc_blocks = link_auc_device(g, bc, bx, gen)
    assert auc_single > 0.6 and auc_blocks > auc_single - 0.03, (auc_blocks, auc_single)


def test_large_graphs_are_fitted_through_the_block_path_on_one_gpu():
    """``fit_transform`` of the public class on a graph far above ``BLOCK_PATH_MIN_NODES``: the block
    path with the automatic plan; every pair is counted, the result has the API's shape."""
    g = E.barabasi_albert(400_000, 5, 42)
    m = E.Node2VecSkipGramEnsmallen(embedding_size=32, epochs=1, iterations=1, walk_length=32,
                                    window_size=3, verbose=False)
    res = m.fit_transform(g, return_dataframe=False).get_all_node_embedding()
    assert res[0].shape == (400_000, 32) and res[1].shape == (400_000, 32)
    assert np.isfinite(res[0]).all() and np.isfinite(res[1]).all()
    plan = m._model.last_plan  # gn2v_train's own choice
    # rows of 32 floats: 861 fit a workgroup's LDS -> resident cells, 2 parts x 256
    assert (plan["world"], plan["parts"], plan["slices"], plan["stripes"]) == (1, 2, 256, 1)
    assert m.get_last_stats()["pairs"] == 400_000 * (2 * 3 * 32 - 3 * 4)
    init = ops.init_table(400_000, 32, 42, 0, 32 ** -0.5).cpu().numpy()
    assert np.abs(res[0] - init).max() > 1e-3


def test_block_path_on_a_directed_weighted_graph_with_trap_nodes():
    """The public class on a graph above the block-path threshold that is directed, weighted and
    has trap nodes (walks that end early) and nodes nobody points to: every pair of every walk is
    trained exactly once and the tables stay finite."""
    rng = np.random.RandomState(7)
    n, e = 70_000, 280_000
    src = rng.randint(0, n // 2, e)          # only the lower half has out-edges
    dst = rng.randint(0, n, e)
    w = rng.uniform(0.1, 3.0, e)
    g = E.CSRGraph.from_edge_list(src, dst, w, number_of_nodes=n, directed=True)
    kw = dict(embedding_size=16, epochs=2, iterations=2, walk_length=12, window_size=3,
              number_of_negative_samples=4, verbose=False)
    import warnings

    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        m = E.Node2VecSkipGramEnsmallen(**kw)
        res = m.fit_transform(g, return_dataframe=False).get_all_node_embedding()
    assert m._model.last_plan["parts"] == 1 and m._model.last_plan["slices"] == 8
    assert np.isfinite(res[0]).all() and np.isfinite(res[1]).all()
    expected = 0
    wp = m._model.walk_params()
    n_walks = g.get_number_of_unique_source_nodes() * 2
    for epoch in range(2):
        wk = ops.walks(g, wp, 42, epoch, 0, n_walks)
        assert int((wk == -1).sum()) > 0  # trap nodes end walks early
        expected += ops.walk_pairs(wk, 3, 1).shape[0]
    assert m.get_last_stats()["pairs"] == expected


def test_rows_of_256_floats_are_trained_in_resident_cells():
    """Rows of 132-256 floats take the resident kernel's CH = 4 instantiation (workgroups of
    eight waves, cells of 134 rows at d = 256): the default fit of a 200 k-node graph counts every
    pair, stays finite, moves the tables and separates edges from random pairs."""
    from sharded_helpers import link_auc_device

    g = E.barabasi_albert(200_000, 8, 42)
    m = E.models.SkipGram(embedding_size=256, epochs=2, iterations=1, walk_length=64, window_size=4,
                          verbose=False)
    c, x, st = m.fit_transform_device(g)
    assert m.last_plan["slices"] == 256 and m.last_plan["parts"] == 6, m.last_plan
    assert st["pairs"] == 2 * 200_000 * (2 * 4 * 64 - 4 * 5)
    assert bool(torch.isfinite(c).all()) and bool(torch.isfinite(x).all())
    gen = torch.Generator(device="cuda")
    gen.manual_seed(1)
    assert link_auc_device(g, c, x, gen) > 0.75  # two short epochs (0.5 at the start; measured 0.83)

"""CPU tests that pin the oracle (oracle/gn2v_oracle.c).

The reference holds no golden vectors for this path (parity unpinned), so the
oracle is pinned by (a) the published known-answer vector of its RNG, (b) independent
re-derivations -- exact node2vec transition probabilities, torch-autograd gradients of the
negative-sampling loss, the closed-form pair count -- and (c) committed self-goldens that freeze
its outputs between rounds (tests/golden/oracle_karate.npz, made by tests/golden/make_oracle_golden.py).
"""
import os

import numpy as np
import pytest
import torch
from scipy import stats

import embiggen_amd as E
from helpers import GOLDEN, exact_second_order_probs, link_auc, ring_of_cliques
from oracle import oracle as O


def test_splitmix64_known_answers():
    """First three outputs of splitmix64 seeded with 0 (Vigna's published test vector)."""
    L = O.lib()
    assert L.o_draw(0, 0) == 0xE220A8397B1DCDAF
    assert L.o_draw(0, 1) == 0x6E789E6AA1B965F4
    assert L.o_draw(0, 2) == 0x06C45D188009454F


def test_walks_follow_edges_and_are_reproducible(karate, karate_oracle):
    wp = O.WalkParams(32, 10, 0.25, 4.0, 100, 0)
    w = O.walks(karate_oracle, wp, 42, 0, 0, 340)
    assert w.shape == (340, 32) and w.dtype == np.uint32
    rp, ci = karate.row_ptr.astype(np.int64), karate.col_idx
    for b in range(340):
        assert w[b, 0] == b % 34  # walk_id = iteration * n_sources + source
        for t in range(31):
            assert w[b, t + 1] in ci[rp[w[b, t]]:rp[w[b, t] + 1]]
    # same (seed, epoch) -> same walks, whatever the batch split
    again = np.concatenate([O.walks(karate_oracle, wp, 42, 0, 0, 100),
                            O.walks(karate_oracle, wp, 42, 0, 100, 240)])
    assert np.array_equal(w, again)
    assert not np.array_equal(w, O.walks(karate_oracle, wp, 42, 1, 0, 340))
    assert not np.array_equal(w, O.walks(karate_oracle, wp, 43, 0, 0, 340))


@pytest.mark.parametrize("rw,ew", [(1.0, 1.0), (0.25, 4.0), (2.0, 0.5), (4.0, 0.25)])
def test_second_order_transition_distribution(karate, karate_oracle, rw, ew):
    """Empirical (prev, cur) -> next frequencies match the exact node2vec distribution."""
    wp = O.WalkParams(12, 1, rw, ew, 100, 0)
    w = O.walks(karate_oracle, wp, 7, 0, 0, 34 * 3000).astype(np.int64)
    prev = w[:, :-2].ravel()
    cur = w[:, 1:-1].ravel()
    nxt = w[:, 2:].ravel()
    key = prev * 34 + cur
    checked, pvals = 0, []
    for k in np.unique(key):
        sel = key == k
        if sel.sum() < 2000:
            continue
        p, c = divmod(int(k), 34)
        neigh, probs = exact_second_order_probs(karate, p, c, rw, ew)
        counts = np.array([(nxt[sel] == x).sum() for x in neigh], dtype=np.float64)
        assert counts.sum() == sel.sum()
        expected = probs * counts.sum()
        if (expected < 5).any():
            continue
        pvals.append(stats.chisquare(counts, expected).pvalue)
        checked += 1
    assert checked >= 20
    assert min(pvals) > 1e-3 / checked, (min(pvals), checked)


def test_return_apart_on_a_directed_graph_where_the_way_back_is_no_edge():
    """return_weight above every other weight: the previous node is proposed on its own
    (walk_consts.apart) and must be turned down when cur -> prev is not an edge.  Directed graph
    0 -> {1, 2}, 1 -> {2, 3, 4}: from (prev 0, cur 1) the walk continues to 2 (adjacent to 0:
    weight 1), 3 or 4 (weight explore_weight), never back to 0."""
    src = np.array([0, 0, 1, 1, 1, 2, 3, 4])
    dst = np.array([1, 2, 2, 3, 4, 0, 0, 0])
    g = E.CSRGraph.from_edge_list(src, dst, number_of_nodes=5, directed=True)
    og = O.OracleGraph(g.row_ptr, g.col_idx)
    rw, ew = 4.0, 0.25
    w = O.walks(og, O.WalkParams(3, 1, rw, ew, 100, 0), 5, 0, 0, 5 * 40000)
    sel = (w[:, 0] == 0) & (w[:, 1] == 1)
    assert sel.sum() > 10000 and not (w[sel, 2] == 0).any()
    counts = np.array([(w[sel, 2] == x).sum() for x in (2, 3, 4)], dtype=np.float64)
    want = np.array([1.0, ew, ew]) / (1.0 + 2 * ew)
    assert stats.chisquare(counts, want * counts.sum()).pvalue > 1e-4
    # and with the way back present (undirected): prev takes return_weight
    g2 = E.CSRGraph.from_edge_list(src[:5], dst[:5], number_of_nodes=5)
    og2 = O.OracleGraph(g2.row_ptr, g2.col_idx)
    w = O.walks(og2, O.WalkParams(3, 1, rw, ew, 100, 0), 5, 0, 0, 5 * 40000)
    sel = (w[:, 0] == 0) & (w[:, 1] == 1)
    counts = np.array([(w[sel, 2] == x).sum() for x in (0, 2, 3, 4)], dtype=np.float64)
    want = np.array([rw, 1.0, ew, ew]) / (rw + 1.0 + 2 * ew)
    assert stats.chisquare(counts, want * counts.sum()).pvalue > 1e-4


def test_extreme_weights_use_exact_fallback(karate, karate_oracle):
    """return_weight >> 1 with explore tiny: acceptance ~1e-4, the rejection loop gives up after
    128 trials and the exact scan must still sample the right distribution."""
    rw, ew = 1.0, 1e-4
    wp = O.WalkParams(3, 1, rw, ew, 100, 0)
    w = O.walks(karate_oracle, wp, 3, 0, 0, 34 * 4000).astype(np.int64)
    sel = (w[:, 0] == 0) & (w[:, 1] == 1)
    neigh, probs = exact_second_order_probs(karate, 0, 1, rw, ew)
    counts = np.array([(w[sel, 2] == x).sum() for x in neigh], dtype=np.float64)
    big = probs * counts.sum() >= 5
    chi = stats.chisquare(counts[big], probs[big] / probs[big].sum() * counts[big].sum())
    assert chi.pvalue > 1e-4


def test_weighted_first_order_distribution():
    src = np.array([0, 0, 0, 1, 2])
    dst = np.array([1, 2, 3, 2, 3])
    wts = np.array([1.0, 2.0, 5.0, 1.0, 1.0])
    g = E.CSRGraph.from_edge_list(src, dst, wts, number_of_nodes=4)
    og = O.OracleGraph(g.row_ptr, g.col_idx, g.cumw)
    w = O.walks(og, O.WalkParams(2, 1, 1.0, 1.0, 100, 0), 1, 0, 0, 4 * 20000)
    from0 = w[w[:, 0] == 0, 1]
    counts = np.array([(from0 == x).sum() for x in (1, 2, 3)], dtype=np.float64)
    assert stats.chisquare(counts, np.array([1, 2, 5]) / 8 * counts.sum()).pvalue > 1e-4


def test_trap_nodes_end_walks_with_sentinel():
    # directed path 0 -> 1 -> 2, node 2 is a trap; node 3 isolated (never a source)
    g = E.CSRGraph.from_edge_list([0, 1], [1, 2], number_of_nodes=4, directed=True)
    og = O.OracleGraph(g.row_ptr, g.col_idx)
    assert list(g.sources) == [0, 1]
    w = O.walks(og, O.WalkParams(5, 1, 1.0, 1.0, 100, 0), 1, 0, 0, 2, sources=g.sources)
    S = O.SENTINEL
    assert w.tolist() == [[0, 1, 2, S, S], [1, 2, S, S, S]]


def test_ba_generator_shape():
    n, m = 20000, 5
    s, d = O.ba_edges(n, m, 42)
    assert len(s) == (n - 1) * m and (s > d).all()
    assert np.array_equal(s, np.repeat(np.arange(1, n), m).astype(np.uint32))
    s2, d2 = O.ba_edges(n, m, 42)
    assert np.array_equal(d, d2) and not np.array_equal(d, O.ba_edges(n, m, 43)[1])
    g = E.CSRGraph.from_edge_list(s, d, number_of_nodes=n)
    deg = g.get_node_degrees()
    assert abs(deg.mean() - 2 * m) < 0.2 and deg.min() >= 1
    # preferential attachment: heavy tail, early nodes are the hubs
    assert deg.max() > 40 * m / 2 and deg[:100].mean() > 10 * deg[-1000:].mean()
    tail = np.sort(deg)[::-1]
    assert tail[10] > 5 * np.median(deg)


def _loss_sgns(u, rows, labels, clip):
    dots = torch.clamp(rows @ u, -clip, clip)
    return -(labels * torch.nn.functional.logsigmoid(dots)
             + (1 - labels) * torch.nn.functional.logsigmoid(-dots)).sum()


def test_sgns_update_is_the_loss_gradient(karate_oracle):
    """One centre with distinct sample rows: the oracle's update == -lr * autograd gradient of
    -log s(u.v+) - sum log s(-u.v-)  (NCE/negative-sampling loss of skipgram.py:45-50)."""
    d, k, w, lr = 8, 3, 1, 0.05
    tp = O.TrainParams(0, d, d, 1, k, w, lr, 1.0, 6.0, 0, 0.5)
    central = O.init_table(34, d, d, 1, 0, 0.5)
    contextual = O.init_table(34, d, d, 1, 1, 0.5)
    c0, x0 = central.copy(), contextual.copy()
    walk = np.array([[5, 9]], dtype=np.uint32)  # centre 5 -> ctx 9 and centre 9 -> ctx 5
    neg = np.zeros((1, 2, 2 * w, k), dtype=np.uint32)
    neg[0, 0, 1] = [11, 12, 13]  # centre 0 only has the right-hand slot (slot index w)
    neg[0, 1, 0] = [20, 21, 22]
    O.train_walks(karate_oracle, tp, walk, 0, 0, 0, lr, central, contextual, neg_override=neg)

    u = torch.tensor(c0[5], requires_grad=True)
    rows = torch.tensor(x0[[9, 11, 12, 13]], requires_grad=True)
    labels = torch.tensor([1.0, 0, 0, 0])
    _loss_sgns(u, rows, labels, 6.0).backward()
    assert np.allclose(central[5], c0[5] - lr * u.grad.numpy(), atol=1e-6)
    assert np.allclose(contextual[[9, 11, 12, 13]], x0[[9, 11, 12, 13]] - lr * rows.grad.numpy(),
                       atol=1e-6)
    u = torch.tensor(c0[9], requires_grad=True)
    rows = torch.tensor(x0[[5, 20, 21, 22]], requires_grad=True)
    _loss_sgns(u, rows, labels, 6.0).backward()
    assert np.allclose(central[9], c0[9] - lr * u.grad.numpy(), atol=1e-6)
    untouched = np.setdiff1d(np.arange(34), [5, 9, 11, 12, 13, 20, 21, 22])
    assert np.array_equal(contextual[untouched], x0[untouched])


def test_cbow_update_is_the_loss_gradient(karate_oracle):
    d, k, w, lr = 8, 3, 2, 0.05
    tp = O.TrainParams(1, d, d, 1, k, w, lr, 1.0, 6.0, 0, 0.5)
    central = O.init_table(34, d, d, 2, 0, 0.5)
    contextual = O.init_table(34, d, d, 2, 1, 0.5)
    c0, x0 = central.copy(), contextual.copy()
    walk = np.array([[1, 2, 3, O.SENTINEL, O.SENTINEL]], dtype=np.uint32)
    neg = np.zeros((1, 5, k), dtype=np.uint32)
    neg[0, 0] = [10, 11, 12]
    neg[0, 1] = [13, 14, 15]
    neg[0, 2] = [16, 17, 18]
    O.train_walks(karate_oracle, tp, walk, 0, 0, 0, lr, central, contextual, neg_override=neg)
    # first centre (node 1): contexts {2, 3}, targets {1, 10, 11, 12}; later centres touch
    # other target rows, so these four rows only saw the first centre
    ctx = torch.tensor(x0[[2, 3]], requires_grad=True)
    tg = torch.tensor(c0[[1, 10, 11, 12]], requires_grad=True)
    labels = torch.tensor([1.0, 0, 0, 0])
    _loss_sgns(ctx.mean(0), tg, labels, 6.0).backward()
    assert np.allclose(central[[10, 11, 12]], c0[[10, 11, 12]] - lr * tg.grad.numpy()[1:],
                       atol=1e-6)


def test_pair_count_and_window_trimming(karate_oracle):
    L, w = 16, 3
    wp = O.WalkParams(L, 2, 1.0, 1.0, 100, 0)
    tp = O.TrainParams(0, 4, 4, 1, 2, w, 0.01, 0.9, 6.0, 1, 0.5)
    _, _, pairs = O.fit(karate_oracle, wp, tp, 5)
    assert pairs == 34 * 2 * (2 * w * L - w * (w + 1))


def test_min_distance_selects_walklets_pairs(karate_oracle):
    """min_dist = window = s keeps only the (centre, context) pairs exactly s steps apart
    (Walklets scale s): pair count 2 * (L - s) per walk, and only those rows move."""
    L, s = 12, 3
    wp = O.WalkParams(L, 1, 1.0, 1.0, 100, 0)
    tp = O.TrainParams(0, 4, 4, 1, 2, s, 0.01, 0.9, 6.0, 1, 0.5, s)
    _, _, pairs = O.fit(karate_oracle, wp, tp, 5)
    assert pairs == 34 * 2 * (L - s)
    # one walk, explicit check of which contextual rows are touched (no negatives)
    tp0 = O.TrainParams(0, 4, 4, 1, 0, 2, 0.05, 1.0, 6.0, 0, 0.5, 2)
    c = O.init_table(34, 4, 4, 1, 0, 0.5)
    x = O.init_table(34, 4, 4, 1, 1, 0.5)
    x0 = x.copy()
    walk = np.array([[0, 1, 2, O.SENTINEL]], dtype=np.uint32)  # only pair at distance 2: (0, 2)
    O.train_walks(karate_oracle, tp0, walk, 0, 0, 0, 0.05, c, x)
    changed = np.flatnonzero(np.abs(x - x0).max(1) > 0)
    assert changed.tolist() == [0, 2]


def test_learning_rate_decay_and_epochs(karate_oracle):
    wp = O.WalkParams(8, 1, 1.0, 1.0, 100, 0)
    base = O.TrainParams(0, 4, 4, 2, 2, 2, 0.05, 0.5, 6.0, 1, 0.5)
    c2, x2, _ = O.fit(karate_oracle, wp, base, 9)
    # replay by hand: epoch 0 at lr, epoch 1 at lr * decay
    c = O.init_table(34, 4, 4, 9, 0, 0.5)
    x = O.init_table(34, 4, 4, 9, 1, 0.5)
    for e, lr in ((0, 0.05), (1, np.float32(0.05) * np.float32(0.5))):
        w = O.walks(karate_oracle, wp, 9, e, 0, 34)
        O.train_walks(karate_oracle, base, w, 9, e, 0, float(lr), c, x)
    assert np.array_equal(c, c2) and np.array_equal(x, x2)


def test_window_batch_matches_sequence_contract(karate_oracle):
    """contexts [n, 2w], words [n], n = walks * (walk_length - 2w)
    (node2vec_sequence.py:115-128)."""
    w = O.walks(karate_oracle, O.WalkParams(10, 1, 1.0, 1.0, 100, 0), 3, 0, 0, 34)
    contexts, words = O.window_batch(w, 2)
    assert contexts.shape == (34 * 6, 4) and words.shape == (34 * 6,)
    assert contexts.dtype == np.int32 and words.dtype == np.int32
    assert words[0] == w[0, 2] and contexts[0].tolist() == [w[0, 0], w[0, 1], w[0, 3], w[0, 4]]
    assert words[7] == w[1, 3]


def test_training_learns_communities():
    """Quality sanity: embeddings of a ring of cliques separate edges from non-edges."""
    src, dst, n = ring_of_cliques(8, 8)
    g = E.CSRGraph.from_edge_list(src, dst, number_of_nodes=n)
    og = O.OracleGraph(g.row_ptr, g.col_idx)
    wp = O.WalkParams(32, 4, 1.0, 1.0, 100, 0)
    tp = O.TrainParams(0, 16, 16, 5, 5, 4, 0.025, 0.9, 6.0, 1, 16 ** -0.5)
    c, x, _ = O.fit(og, wp, tp, 42)
    assert link_auc(g, c, x) > 0.9
    tp.model = 1
    c, x, _ = O.fit(og, wp, tp, 42)
    assert np.isfinite(c).all() and link_auc(g, x, c) > 0.85


def test_hogwild_threads_are_statistically_equivalent(karate, karate_oracle):
    """The OpenMP Hogwild variant (the CPU baseline) races by design: same work, same quality,
    different trajectory."""
    wp = O.WalkParams(16, 4, 0.25, 4.0, 100, 0)
    tp = O.TrainParams(0, 8, 8, 4, 5, 3, 0.025, 0.9, 6.0, 1, 8 ** -0.5)
    c1, x1, p1 = O.fit(karate_oracle, wp, tp, 42, threads=1)
    c4, x4, p4 = O.fit(karate_oracle, wp, tp, 42, threads=4)
    assert p1 == p4 and np.isfinite(c4).all() and np.isfinite(x4).all()
    assert abs(link_auc(karate, c1, x1) - link_auc(karate, c4, x4)) < 0.05


def test_self_golden(karate_oracle):
    """Freeze the oracle's own outputs (regression pin, not reference-derived)."""
    gold = np.load(os.path.join(GOLDEN, "oracle_karate.npz"))
    wp = O.WalkParams(16, 2, 0.25, 4.0, 100, 0)
    assert np.array_equal(O.walks(karate_oracle, wp, 42, 0, 0, 68), gold["walks"])
    for model, key in ((0, "sgns"), (1, "cbow")):
        tp = O.TrainParams(model, 8, 8, 2, 4, 3, 0.01, 0.9, 6.0, 1, 8 ** -0.5)
        c, x, pairs = O.fit(karate_oracle, wp, tp, 42)
        assert pairs == int(gold[f"{key}_pairs"])
        assert np.allclose(c, gold[f"{key}_central"], atol=1e-6)
        assert np.allclose(x, gold[f"{key}_contextual"], atol=1e-6)
    s, d = O.ba_edges(500, 3, 42)
    assert np.array_equal(d, gold["ba_dst"])


def test_self_golden_of_the_block_schedule(karate_oracle):
    """The restated block-partitioned schedule, frozen (regression pin between rounds, not
    reference-derived; made by tests/golden/make_oracle_golden.py): extraction + sort, alias
    tables with two hot rows per cell, one round over every part."""
    gold = np.load(os.path.join(GOLDEN, "oracle_blocks.npz"))
    walks = np.load(os.path.join(GOLDEN, "oracle_karate.npz"))["walks"]
    plan = O.block_plan(34, 2, 1, 4, 2, 16, 3, 1, 4, hot_rows=2)
    alias, cell_rows, hub_bits, hot_list, hot_slot = O.block_alias(karate_oracle, 4, 2, 2)
    words, offsets = O.block_extract(karate_oracle, plan, walks, 42, 0, 0, hub_bits=hub_bits)
    assert O.block_unpack(words, plan)[3].any()  # some context rows are hot
    for name, got in (("words", words), ("offsets", offsets), ("alias", alias),
                      ("cell_rows", cell_rows), ("hub_bits", hub_bits), ("hot_list", hot_list),
                      ("hot_slot", hot_slot)):
        assert np.array_equal(got, gold[name]), name
    tp = O.TrainParams(0, 8, 8, 1, 4, 3, 0.01, 0.9, 6.0, 1, 8 ** -0.5)
    central = O.init_table_rows(17, 8, 8, 42, 0, 8 ** -0.5, 1, 2)
    for part in range(4):
        x = O.init_table_rows((34 - part + 3) // 4, 8, 8, 42, 1, 8 ** -0.5, part, 4)
        O.block_step(karate_oracle, tp, plan, words, offsets, alias, cell_rows, central, x, 7,
                     part, 42, 0, 0.05)
        assert np.allclose(x, gold[f"part{part}"], atol=1e-6)
    assert np.allclose(central, gold["central"], atol=1e-6)
    # the same walks under the placement of round 3 (round 5: cells change every round)
    place, inv = O.block_placement(34, 1, 42, 3)
    rplan = O.block_plan(34, 1, 0, 1, 17, 16, 3, 1, 8)
    ralias, rcell_rows = O.block_alias(karate_oracle, 1, 17, 0, inv=inv)[:2]
    rwords, roffsets = O.block_extract(karate_oracle, rplan, walks, 42, 0, 0, place=place)
    for name, got in (("placed_place", place), ("placed_inv", inv), ("placed_words", rwords),
                      ("placed_offsets", roffsets), ("placed_alias", ralias)):
        assert np.array_equal(got, gold[name]), name
    rc = O.init_table(34, 8, 8, 42, 0, 8 ** -0.5)
    rx = O.init_table(34, 8, 8, 42, 1, 8 ** -0.5)
    O.block_step(karate_oracle, tp, rplan, rwords, roffsets, ralias, rcell_rows, rc, rx, 3, 0, 42,
                 0, 0.05, inv=inv, natural=True)
    assert np.allclose(rc, gold["placed_central"], atol=1e-6)
    assert np.allclose(rx, gold["placed_contextual"], atol=1e-6)


def test_oracle_is_clean_under_address_and_ub_sanitizers():
    """GPU sanitizers are not available on the pool; the CPU restatement of every entry point
    (walks incl. weighted / traps / fallback, window, pairs, SkipGram / CBOW fits with all flag
    combinations, pair mode with pools) runs under ASan + UBSan instead."""
    import subprocess

    root = os.path.join(os.path.dirname(GOLDEN), "..", "oracle")
    res = subprocess.run(["make", "-C", root, "sanitize"], capture_output=True, text=True)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-2000:]
    assert "ok " in res.stdout


def test_normalize_by_degree_is_a_destination_degree_weighting(karate):
    """normalize_by_degree (node2vec_skipgram.py:94-96): transition weight divided by the degree of
    the destination -> realised as edge weights 1/deg(dst); first-order frequencies follow it."""
    h = karate.with_degree_normalized_weights()
    deg = karate.get_node_degrees()
    assert np.allclose(h.get_directed_edge_weights(), 1.0 / deg[karate.col_idx])
    og = O.OracleGraph(h.row_ptr, h.col_idx, h.cumw)
    w = O.walks(og, O.WalkParams(2, 1, 1.0, 1.0, 100, 0), 5, 0, 0, 34 * 6000)
    rp = karate.row_ptr.astype(np.int64)
    for u in (0, 33, 2):
        nxt = w[w[:, 0] == u, 1]
        neigh = karate.col_idx[rp[u]:rp[u + 1]]
        counts = np.array([(nxt == x).sum() for x in neigh], dtype=np.float64)
        p = 1.0 / deg[neigh]
        assert stats.chisquare(counts, p / p.sum() * counts.sum()).pvalue > 1e-4


@pytest.mark.parametrize("rw,ew", [(0.25, 4.0), (2.0, 0.5), (1.0, 1.0)])
def test_max_neighbours_above_every_degree_or_none_walks_exactly(karate_oracle, rw, ew):
    """``max_neighbours`` (node2vec_skipgram.py:22,78-81; None = exact) acts on rows LONGER than
    it only: Karate's largest degree is 17, so 100 (the default), 17 and 0 (= None at the C
    boundary) give the same walks -- and the default on a graph without hubs is the exact walk."""
    ref = O.walks(karate_oracle, O.WalkParams(12, 1, rw, ew, 0, 0), 7, 0, 0, 34 * 40)
    for m in (100, 17):
        assert np.array_equal(O.walks(karate_oracle, O.WalkParams(12, 1, rw, ew, m, 0), 7, 0, 0,
                                      34 * 40), ref)
    assert not np.array_equal(O.walks(karate_oracle, O.WalkParams(12, 1, rw, ew, 16, 0), 7, 0, 0,
                                      34 * 40), ref)


@pytest.mark.parametrize("rw,ew", [(0.25, 4.0), (2.0, 0.5), (1.0, 1.0)])
@pytest.mark.parametrize("max_neighbours", [10, 3])
def test_max_neighbours_walks_hubs_over_a_sub_sample(karate, karate_oracle, max_neighbours, rw, ew):
    """"Number of maximum neighbours to consider when using approximated walks ... mainly useful
    for graphs containing nodes with high degrees" (node2vec_skipgram.py:78-81; the smoke
    configuration sets 10, node2vec.py:79-87).  Restated (oracle row_view): a step out of a node
    of degree > max_neighbours chooses among a sub-sample of max_neighbours of its edges -- one
    per bucket of the row, drawn afresh at every visit -- by the exact node2vec law on them.
    Checked against an independent ENUMERATION of every possible sub-sample
    (helpers.exact_sub_sampled_probs): chi-square of the (prev, cur) -> next frequencies for the
    hubs, the exact law for every other node; both envelopes of the rejection sampler (return
    apart and not) and first-order walks."""
    from helpers import exact_sub_sampled_probs

    deg = np.diff(karate.row_ptr.astype(np.int64))
    assert deg.max() > max_neighbours
    w = O.walks(karate_oracle, O.WalkParams(12, 1, rw, ew, max_neighbours, 0), 7, 0, 0,
                34 * 3000).astype(np.int64)
    prev, cur, nxt = w[:, :-2].ravel(), w[:, 1:-1].ravel(), w[:, 2:].ravel()
    key = prev * 34 + cur
    pvals = {True: [], False: []}
    for k in np.unique(key):
        p, c = divmod(int(k), 34)
        sel = key == k
        hub = deg[c] > max_neighbours
        if sel.sum() < 2000 or (hub and max_neighbours == 3 and deg[c] > 12):
            continue  # (3 buckets of a row of 16-17 edges: 200 sub-samples: fine; keep it quick)
        neigh, probs = (exact_sub_sampled_probs(karate, p, c, rw, ew, max_neighbours) if hub
                        else exact_second_order_probs(karate, p, c, rw, ew))
        counts = np.array([(nxt[sel] == x).sum() for x in neigh], dtype=np.float64)
        assert counts.sum() == sel.sum()
        keep = probs * counts.sum() >= 5
        if keep.sum() < 2:
            continue
        pvals[bool(hub)].append(stats.chisquare(
            counts[keep], probs[keep] / probs[keep].sum() * counts[keep].sum()).pvalue)
    n = len(pvals[True]) + len(pvals[False])
    assert len(pvals[True]) >= 5 and len(pvals[False]) >= 5, {k: len(v) for k, v in pvals.items()}
    assert min(pvals[True] + pvals[False]) > 1e-3 / n, (pvals,)
    # and the sub-sampled law is NOT the exact one (the test can tell them apart) when the
    # weights differ inside a row
    if (rw, ew) != (1.0, 1.0) and max_neighbours == 3:
        far = []
        for k in np.unique(key):
            p, c = divmod(int(k), 34)
            sel = key == k
            if deg[c] <= 12 and deg[c] > 3 and sel.sum() >= 4000:
                neigh, probs = exact_second_order_probs(karate, p, c, rw, ew)
                counts = np.array([(nxt[sel] == x).sum() for x in neigh], dtype=np.float64)
                keep = probs * counts.sum() >= 5
                far.append(stats.chisquare(counts[keep], probs[keep] / probs[keep].sum()
                                           * counts[keep].sum()).pvalue)
        assert far and min(far) < 1e-6, far


def test_tuned_oracle_build_is_the_same_algorithm(karate_oracle):
    """oracle/libgn2v_oracle_fast.so (``-ffast-math -DO_FAST``: what ``bench.py``'s
    ``cpu_baseline`` times) is the strict oracle's source with reassociated sums and prefetch
    hints: same updates to 1e-5, SkipGram and CBOW.  It is never the checker: every parity test
    uses the strict build."""
    wp = O.WalkParams(16, 2, 0.25, 4.0, 100, 0)
    walks = O.walks(karate_oracle, wp, 42, 0, 0, 68)
    for model in (0, 1):
        tp = O.TrainParams(model, 8, 8, 1, 4, 3, 0.01, 0.9, 6.0, 1, 8 ** -0.5)
        out = []
        for fast in (False, True):
            c = O.init_table(34, 8, 8, 42, 0, 8 ** -0.5)
            x = O.init_table(34, 8, 8, 42, 1, 8 ** -0.5)
            O.train_walks(karate_oracle, tp, walks, 42, 0, 0, 0.05, c, x, fast=fast)
            out.append((c, x))
        assert np.abs(out[0][0] - out[1][0]).max() < 1e-5
        assert np.abs(out[0][1] - out[1][1]).max() < 1e-5
        assert np.abs(out[0][0] - O.init_table(34, 8, 8, 42, 0, 8 ** -0.5)).max() > 1e-3

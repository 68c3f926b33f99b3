"""Property tests of the oracle on random graphs (hypothesis): size-independent invariants of the
walk sampler, the window / pair extraction and the co-occurrence counts."""
import numpy as np
from hypothesis import given, settings
from hypothesis import strategies as st

import embiggen_amd as E
from oracle import oracle as O


@st.composite
def graphs(draw):
    n = draw(st.integers(2, 40))
    e = draw(st.integers(1, 4 * n))
    seed = draw(st.integers(0, 2 ** 31 - 1))
    rng = np.random.RandomState(seed)
    src, dst = rng.randint(0, n, e), rng.randint(0, n, e)
    weights = rng.uniform(0.1, 3.0, e) if draw(st.booleans()) else None
    node_types = rng.randint(0, 3, n).tolist() if draw(st.booleans()) else None
    edge_types = rng.randint(0, 3, e).tolist() if draw(st.booleans()) else None
    return E.CSRGraph.from_edge_list(src, dst, weights, number_of_nodes=n,
                                     directed=draw(st.booleans()), node_types=node_types,
                                     edge_types=edge_types)


weights_st = st.sampled_from([0.25, 0.5, 1.0, 2.0, 4.0])


@settings(max_examples=60, deadline=None)
@given(graphs(), st.integers(2, 30), weights_st, weights_st, weights_st, weights_st,
       st.integers(0, 2 ** 40))
def test_walks_follow_edges_and_end_only_in_traps(g, L, rw, ew, cn, ce, seed):
    og = O.OracleGraph(g.row_ptr, g.col_idx, g.cumw, g.node_type_ids, g.edge_type_ids)
    n_src = g.get_number_of_unique_source_nodes()
    walks = O.walks(og, O.WalkParams(L, 2, rw, ew, 100, 0, cn, ce), seed, 0, 0, 2 * n_src,
                    sources=g.sources)
    rp = g.row_ptr.astype(np.int64)
    starts = g.sources if g.sources is not None else np.arange(g.get_number_of_nodes())
    assert np.array_equal(walks[:, 0], np.tile(starts, 2))
    for wk in walks:
        valid = wk != O.SENTINEL
        k = int(valid.sum())
        assert valid[:k].all() and not valid[k:].any()  # sentinels are a suffix
        for a, b in zip(wk[:k - 1], wk[1:k]):
            assert b in g.col_idx[rp[a]:rp[a + 1]]
        if k < L:  # the walk stopped: its last node has no outgoing edge
            assert rp[wk[k - 1] + 1] == rp[wk[k - 1]]
    again = O.walks(og, O.WalkParams(L, 2, rw, ew, 100, 0, cn, ce), seed, 0, 0, 2 * n_src,
                    sources=g.sources)
    assert np.array_equal(walks, again)


@settings(max_examples=40, deadline=None)
@given(graphs(), st.integers(2, 24), st.integers(1, 6), st.integers(0, 2 ** 40))
def test_pairs_windows_and_cooccurrences_agree(g, L, w, seed):
    og = O.OracleGraph(g.row_ptr, g.col_idx, g.cumw)
    n_src = g.get_number_of_unique_source_nodes()
    walks = O.walks(og, O.WalkParams(L, 1, 1.0, 1.0, 100, 0), seed, 0, 0, n_src, sources=g.sources)
    for md in {1, w}:
        pairs = O.walk_pairs(walks, w, md)
        keys, weights = O.cooc_slots(walks, w, md)
        used = keys != O.COOC_UNUSED
        # one co-occurrence slot per (centre, context) pair, same multiset of pairs
        assert used.sum() == len(pairs)
        from_pairs = (pairs[:, 0].astype(np.uint64) << np.uint64(32)) | pairs[:, 1].astype(np.uint64)
        assert np.array_equal(np.sort(from_pairs), np.sort(keys[used]))
        ukeys, counts = O.cooc_reduce(keys, weights)
        assert counts.sum() == weights[used].sum() and len(np.unique(ukeys)) == len(ukeys)
        # full walks: the count is the closed form of SURVEY 8(a4): sum over positions
        full = (walks != O.SENTINEL).all(axis=1)
        if full.all() and md == 1:
            assert len(pairs) == len(walks) * sum(
                min(i, w) + min(L - 1 - i, w) for i in range(L))
    if L > 2 * w:
        contexts, words = O.window_batch(walks, w)
        assert words.shape == (len(walks) * (L - 2 * w),) and contexts.shape[1] == 2 * w
        assert np.array_equal(words.reshape(len(walks), -1), walks[:, w:L - w].view(np.int32))

"""GloVe restatement in the oracle and the host-side co-occurrence plumbing (CPU).

The reference's walk-based table maps "Node2Vec / DeepWalk / Walklets GloVe" to the ensmallen
wheel (embedders/ensmallen_embedders/node2vec.py:16-26) and holds no vectors for it (parity
unpinned); these tests pin the oracle to the published objective it states and the device-side
tensor plumbing (embiggen_amd/cooccurrence.py, run on CPU tensors here) to the oracle."""
import numpy as np
import pytest
import torch

import embiggen_amd as E
from embiggen_amd import cooccurrence
from helpers import link_auc, ring_of_cliques
from oracle import oracle as O

ONE = 1 << 20


def karate_walks(karate_oracle, L=24, n=68, rw=0.5, ew=2.0):
    return O.walks(karate_oracle, O.WalkParams(L, 2, rw, ew, 100, 0), 7, 0, 0, n)


def naive_counts(walks, window, min_dist=1):
    out = {}
    for wk in walks:
        valid = [int(v) for v in wk if v != 0xFFFFFFFF]
        for i, c in enumerate(valid):
            for j in range(max(0, i - window), min(len(valid), i + window + 1)):
                dist = abs(i - j)
                if dist == 0 or dist < min_dist:
                    continue
                key = (c << 32) | valid[j]
                out[key] = out.get(key, 0) + (ONE + dist // 2) // dist
    return out


@pytest.mark.parametrize("window,min_dist", [(1, 1), (5, 1), (4, 4), (3, 2)])
def test_cooccurrence_counts_match_a_naive_count(karate_oracle, window, min_dist):
    walks = karate_walks(karate_oracle)
    keys, weights = O.cooc_slots(walks, window, min_dist)
    assert len(keys) == walks.size * 2 * window
    ukeys, counts = O.cooc_reduce(keys, weights)
    want = naive_counts(walks, window, min_dist)
    assert dict(zip(ukeys.tolist(), counts.tolist())) == want
    assert (np.diff(ukeys.astype(np.int64)) > 0).all()
    # symmetric window => symmetric matrix
    swapped = ((ukeys & np.uint64(0xFFFFFFFF)) << np.uint64(32)) | (ukeys >> np.uint64(32))
    assert want == {int(k): int(c) for k, c in zip(swapped, counts)}


def test_trapped_walks_only_count_their_valid_prefix():
    g = E.CSRGraph.from_edge_list([0, 1], [1, 2], number_of_nodes=4, directed=True)
    og = O.OracleGraph(g.row_ptr, g.col_idx)
    walks = O.walks(og, O.WalkParams(6, 1, 1.0, 1.0, 100, 0), 1, 0, 0, 2, sources=g.sources)
    assert (walks == 0xFFFFFFFF).any()
    keys, weights = O.cooc_slots(walks, 2)
    ukeys, counts = O.cooc_reduce(keys, weights)
    assert dict(zip(ukeys.tolist(), counts.tolist())) == naive_counts(walks, 2)


def test_entries_are_shuffled_row_records(karate_oracle):
    ukeys, counts = O.cooc_reduce(*O.cooc_slots(karate_walks(karate_oracle), 5))
    rows, cols, logx, fx = O.glove_entries(ukeys, counts, 42, 0.75)
    R = O.GLOVE_RECORD
    assert len(rows) % R == 0 and (rows.reshape(-1, R) == rows.reshape(-1, R)[:, :1]).all()
    used = cols != O.SENTINEL
    # padding only at the end of a record, and only in the last record of a row
    pad = (~used).reshape(-1, R)
    assert (pad[:, 1:] >= pad[:, :-1]).all() and not pad[:, 0].any()
    assert pad.any(axis=1).sum() <= len(np.unique(rows))
    got = (rows[used].astype(np.uint64) << np.uint64(32)) | cols[used].astype(np.uint64)
    assert sorted(got.tolist()) == ukeys.tolist()
    x = np.exp(logx[used].astype(np.float64))
    assert x.max() == pytest.approx(1.0) and (logx <= 0).all() and (x > 0).all()
    assert np.allclose(fx[used], x ** 0.75, rtol=1e-6) and (fx[~used] == 0).all()
    order = np.argsort(got)
    assert np.allclose(x[order], counts / counts.max(), rtol=1e-6)
    # records of one row are scattered, not adjacent
    first = rows.reshape(-1, R)[:, 0]
    assert (first[1:] != first[:-1]).mean() > 0.8
    rows2 = O.glove_entries(ukeys, counts, 43, 0.75)[0]
    assert not np.array_equal(rows, rows2)
    again = O.glove_entries(ukeys, counts, 42, 0.75)
    assert np.array_equal(rows, again[0]) and np.array_equal(cols, again[1])


def test_round_schedule_is_sgd_with_a_row_refreshed_every_four_entries(karate_oracle):
    """The engine's record schedule (o_glove_step_rounds) equals the per-entry loop when every
    round holds one entry, and stays close to it otherwise."""
    ukeys, counts = O.cooc_reduce(*O.cooc_slots(karate_walks(karate_oracle), 3))
    rows, cols, logx, fx = O.glove_entries(ukeys, counts, 1, 0.75)
    d = 8
    make = lambda: [O.init_table(34, d, d, 5, 0, 0.3), O.init_table(34, d, d, 5, 1, 0.3),  # noqa: E731
                    np.zeros(34, np.float32), np.zeros(34, np.float32)]
    # one entry per round: spread every entry to the first slot of its own round of four
    n = len(rows)
    sr, sc = np.repeat(rows, 4), np.full(4 * n, O.SENTINEL, np.uint32)
    sl, sf = np.zeros(4 * n, np.float32), np.zeros(4 * n, np.float32)
    sc[::4], sl[::4], sf[::4] = cols, logx, fx
    a, b = make(), make()
    O.glove_step(rows, cols, logx, fx, *a, d, 0.05)
    O.glove_step_rounds(sr, sc, sl, sf, *b, d, 0.05)
    assert all(np.array_equal(p, q) for p, q in zip(a, b))
    c = make()
    O.glove_step_rounds(rows, cols, logx, fx, *c, d, 0.05)
    assert not np.array_equal(a[0], c[0]) and np.abs(a[0] - c[0]).max() < 0.05


def test_glove_update_is_the_loss_gradient():
    """One entry: the in-place update equals -lr * d(loss)/d(param) of f (u.v + b + b~ - log X)^2 / 2."""
    rng = np.random.RandomState(0)
    d, ld, n = 6, 8, 5
    c0 = np.zeros((n, ld), np.float32)
    x0 = np.zeros((n, ld), np.float32)
    c0[:, :d] = rng.normal(size=(n, d)) * 0.3
    x0[:, :d] = rng.normal(size=(n, d)) * 0.3
    bc0, bx0 = rng.normal(size=n).astype(np.float32), rng.normal(size=n).astype(np.float32)
    rows, cols = np.array([3], np.uint32), np.array([1], np.uint32)
    logx, fx = np.array([-1.7], np.float32), np.array([0.4], np.float32)
    c, x, bc, bx = c0.copy(), x0.copy(), bc0.copy(), bx0.copy()
    O.glove_step(rows, cols, logx, fx, c, x, bc, bx, d, 0.05)
    tc, tx = torch.tensor(c0, requires_grad=True), torch.tensor(x0, requires_grad=True)
    tbc, tbx = torch.tensor(bc0, requires_grad=True), torch.tensor(bx0, requires_grad=True)
    loss = 0.5 * 0.4 * ((tc[3, :d] * tx[1, :d]).sum() + tbc[3] + tbx[1] + 1.7) ** 2
    loss.backward()
    assert np.allclose(c, c0 - 0.05 * tc.grad.numpy(), atol=1e-6)
    assert np.allclose(x, x0 - 0.05 * tx.grad.numpy(), atol=1e-6)
    assert np.allclose(bc, bc0 - 0.05 * tbc.grad.numpy(), atol=1e-6)
    assert np.allclose(bx, bx0 - 0.05 * tbx.grad.numpy(), atol=1e-6)
    assert (c[:, d:] == 0).all() and (x[:, d:] == 0).all()
    assert O.glove_loss(rows, cols, logx, fx, c0, x0, bc0, bx0, d) == pytest.approx(float(loss.detach()), rel=1e-5)


def test_entries_are_applied_sequentially(karate_oracle):
    """Two passes of one entry each == one pass over both (later entries see earlier updates)."""
    ukeys, counts = O.cooc_reduce(*O.cooc_slots(karate_walks(karate_oracle), 3))
    rows, cols, logx, fx = O.glove_entries(ukeys, counts, 1, 0.75)
    d = 8
    init = lambda t: O.init_table(34, d, d, 5, t, 0.3)  # noqa: E731
    a = [init(0), init(1), np.zeros(34, np.float32), np.zeros(34, np.float32)]
    b = [t.copy() for t in a]
    O.glove_step(rows[:40], cols[:40], logx[:40], fx[:40], *a, d, 0.05)
    for e in range(40):
        O.glove_step(rows[e:e + 1], cols[e:e + 1], logx[e:e + 1], fx[e:e + 1], *b, d, 0.05)
    assert all(np.array_equal(p, q) for p, q in zip(a, b))


def test_glove_training_learns_communities():
    src, dst, n = ring_of_cliques(8, 8)
    g = E.CSRGraph.from_edge_list(src, dst, number_of_nodes=n)
    og = O.OracleGraph(g.row_ptr, g.col_idx)
    walks = O.walks(og, O.WalkParams(64, 1, 1.0, 1.0, 100, 0), 42, 0, 0, n)
    rows, cols, logx, fx = O.glove_entries(*O.cooc_reduce(*O.cooc_slots(walks, 4)), 42, 0.75)
    d = 16
    c, x = O.init_table(n, d, d, 42, 0, d ** -0.5), O.init_table(n, d, d, 42, 1, d ** -0.5)
    bc, bx = np.zeros(n, np.float32), np.zeros(n, np.float32)
    first = O.glove_loss(rows, cols, logx, fx, c, x, bc, bx, d)
    lr = np.float32(0.05)
    for _ in range(100):  # the DeepWalk GloVe wrapper's schedule (deepwalk_glove.py:12-18)
        O.glove_step(rows, cols, logx, fx, c, x, bc, bx, d, float(lr))
        lr = np.float32(lr * np.float32(0.99))
    last = O.glove_loss(rows, cols, logx, fx, c, x, bc, bx, d)
    assert np.isfinite(c).all() and last < 0.2 * first
    assert link_auc(g, c, x) > 0.9


def test_tensor_plumbing_equals_the_oracle(karate_oracle):
    """embiggen_amd.cooccurrence (torch ops; CPU tensors here, HBM tensors in the product)."""
    walks = karate_walks(karate_oracle, n=102)
    keys, weights = O.cooc_slots(walks, 4)
    want_keys, want_counts = O.cooc_reduce(keys, weights)
    tk, tw = torch.from_numpy(keys.view(np.int64)), torch.from_numpy(weights.view(np.int64))
    got_keys, got_counts = cooccurrence.reduce_slots(tk, tw)
    assert np.array_equal(got_keys.numpy().view(np.uint64), want_keys)
    assert np.array_equal(got_counts.numpy().view(np.uint64), want_counts)
    # merging batches gives the same sums as reducing everything at once
    half = len(keys) // 2
    merged = cooccurrence.merge(cooccurrence.merge(None, cooccurrence.reduce_slots(tk[:half], tw[:half])),
                                cooccurrence.reduce_slots(tk[half:], tw[half:]))
    assert torch.equal(merged[0], got_keys) and torch.equal(merged[1], got_counts)
    acc = cooccurrence.Accumulator()
    for lo in range(0, len(keys), len(keys) // 7):
        acc.add(cooccurrence.reduce_slots(tk[lo:lo + len(keys) // 7], tw[lo:lo + len(keys) // 7]))
    total = acc.result()
    assert torch.equal(total[0], got_keys) and torch.equal(total[1], got_counts)
    assert cooccurrence.Accumulator().result() is None
    none = cooccurrence.reduce_slots(torch.full((6,), cooccurrence.UNUSED, dtype=torch.int64),
                                     torch.zeros(6, dtype=torch.int64))
    assert none[0].numel() == 0 and none[1].numel() == 0
    assert cooccurrence.reduce_slots(tk[:0], tw[:0])[0].numel() == 0
    one = cooccurrence.reduce_slots(torch.tensor([5, 5, 5]), torch.tensor([1, 2, 3]))
    assert one[0].tolist() == [5] and one[1].tolist() == [6]
    for seed in (0, 42, 2 ** 40 + 3):
        rows, cols, logx, fx = cooccurrence.entries(got_keys, got_counts, seed, 0.75)
        orows, ocols, ologx, ofx = O.glove_entries(want_keys, want_counts, seed, 0.75)
        assert np.array_equal(rows.numpy().view(np.uint32), orows)
        assert np.array_equal(cols.numpy().view(np.uint32), ocols)
        assert np.allclose(logx.numpy(), ologx, rtol=1e-6, atol=1e-7)
        assert np.allclose(fx.numpy(), ofx, rtol=1e-6)
    assert cooccurrence.mix64_int(12345) == O.mix64(12345)
    z = torch.tensor([0, 1, -1, 2 ** 62, -(2 ** 63)], dtype=torch.int64)
    assert [v & (2 ** 64 - 1) for v in cooccurrence.mix64_tensor(z).tolist()] == [
        O.mix64(v & (2 ** 64 - 1)) for v in z.tolist()]


def test_merging_runs_longer_than_one_sort_allows(monkeypatch):
    """torch.sort refuses dimensions above INT_MAX: reduced runs whose union is longer (seen on a
    default GloVe fit of a 1 M-node graph) are cut at a pivot key and merged by halves."""
    import torch

    from embiggen_amd import cooccurrence as CO

    rng = np.random.RandomState(0)

    def run(n):
        k = np.unique(rng.randint(0, 10 ** 6, n))
        return (torch.from_numpy(k.astype(np.int64)),
                torch.from_numpy(rng.randint(1, 100, len(k)).astype(np.int64)))

    a, b = run(5000), run(7000)
    want = CO.merge(a, b)
    monkeypatch.setattr(CO, "SORT_LIMIT", 1000)
    got = CO.merge(a, b)
    assert torch.equal(want[0], got[0]) and torch.equal(want[1], got[1])
    acc = CO.Accumulator()
    total = {}
    for _ in range(6):
        r = run(3000)
        acc.add(r)
        for k, w in zip(r[0].tolist(), r[1].tolist()):
            total[k] = total.get(k, 0) + w
    keys, counts = acc.result()
    assert keys.tolist() == sorted(total) and counts.tolist() == [total[k] for k in sorted(total)]


def test_ordering_more_entries_than_one_sort_allows(monkeypatch):
    """The training order of `entries` (row, hash of key) when the entry set is longer than one
    sort takes (a default GloVe fit of a 1 M-node graph holds > 2^31 distinct pairs): pieces cut
    where a row starts give the same slots as one sort."""
    import torch

    from embiggen_amd import cooccurrence as CO

    rng = np.random.RandomState(3)
    rows = np.sort(rng.randint(0, 300, 6000)).astype(np.int64)
    keys = np.unique((rows << 32) | rng.randint(0, 5000, 6000))
    tk = torch.from_numpy(keys)
    tc = torch.from_numpy(rng.randint(1, 1 << 22, len(keys)).astype(np.int64))
    want = CO.entries(tk, tc, 42, 0.75)
    monkeypatch.setattr(CO, "SORT_LIMIT", 1000)  # six ranges of rows; the 541 records fit one sort
    got = CO.entries(tk, tc, 42, 0.75)
    for w, g in zip(want, got):
        assert torch.equal(w, g)
    # a row that alone exceeds the limit cannot be cut
    monkeypatch.setattr(CO, "SORT_LIMIT", 8)
    with pytest.raises(RuntimeError, match="one row holds"):
        CO.entries(tk, tc, 42, 0.75)

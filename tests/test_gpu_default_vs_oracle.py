"""The schedule that SHIPS (parallel block path: racing stores on ordinary contextual rows, LDS
copies of the hot ones, one store or atomics on the central rows) against the oracle on identical
seeded walks -- the closest runnable form of the north star's "cosine similarities within 1e-2 on
identical seeded walks" (the reference's CPU path is racy as well, so two parallel fits agree
statistically, not element by element; the deterministic schedule is held to 1e-5 elsewhere).

* config 2's shape (BA 2 708 / 5.4 k, d = 128, p = q = 1) and config 3's (BA 169 343 / 1.17 M,
  p = 0.5, q = 2): one walk per node, one epoch; the default GPU fit and the oracle's Hogwild
  trainer (all host threads) on the same walks; cosine similarities of edges and of random pairs
  under both fits: rank correlation and mean absolute difference, gated.
* config 2's shape again against the oracle's SEQUENTIAL restatement of the block schedule -- the
  same pairs, the same negatives, one at a time: what is left is what parallel execution does.
* the single-run store on central rows: how many updates it loses, counted.
"""
import os

import numpy as np
import pytest
import torch

import embiggen_amd as E
from embiggen_amd import models, ops
from embiggen_amd.distributed import BlockPartitionedTrainer, LoopbackComm
from oracle import oracle as O
from sharded_helpers import OracleBlockBackend

pytestmark = pytest.mark.gpu


def _pairs(g, n_each, seed):
    """(src, dst) of n_each random directed edges followed by n_each random node pairs."""
    rng = np.random.RandomState(seed)
    n = g.get_number_of_nodes()
    e = rng.randint(0, len(g.col_idx), n_each)
    src = np.searchsorted(g.row_ptr, e, side="right") - 1
    dst = g.col_idx[e].astype(np.int64)
    return (np.concatenate([src, rng.randint(0, n, n_each)]),
            np.concatenate([dst, rng.randint(0, n, n_each)]))


def _cos(table, u, v):
    a, b = table[u].astype(np.float64), table[v].astype(np.float64)
    return (a * b).sum(1) / np.maximum(np.linalg.norm(a, axis=1) * np.linalg.norm(b, axis=1), 1e-6)


def _agreement(g, got, want, n_each=100_000):
    from scipy import stats

    u, v = _pairs(g, n_each, 7)
    a, b = _cos(got, u, v), _cos(want, u, v)
    auc = lambda s: float((stats.rankdata(s)[:n_each].sum() - n_each * (n_each + 1) / 2)  # noqa: E731
                          / (n_each * n_each))
    return dict(spearman=float(stats.spearmanr(a, b)[0]), mean_abs=float(np.abs(a - b).mean()),
                auc_got=auc(a), auc_want=auc(b))


SHAPES = {
    # name: (nodes, m, walk_length, return_weight, explore_weight)
    "config2_cora_shape": (2_708, 2, 128, 1.0, 1.0),
    "config3_arxiv_shape": (169_343, 7, 64, 2.0, 0.5),
}


@pytest.mark.timeout(900)
@pytest.mark.parametrize("shape", list(SHAPES))
def test_default_fit_agrees_with_the_oracles_hogwild_fit_on_the_same_walks(shape):
    """One walk per node, one epoch of the reference defaults at d = 128 (window 5, 10 negatives,
    lr 0.01): the two parallel fits must rank the cosine similarities of 10^5 edges + 10^5
    random pairs alike (Spearman >= 0.9), differ little in value, and separate edges from random
    pairs equally well.  Different negatives (the block path draws them inside the context's
    cell), different races: what agrees is what the walks teach."""
    nodes, m, L, rw, ew = SHAPES[shape]
    g = E.barabasi_albert(nodes, m, 42)
    host = g  # row_ptr / col_idx come to the host on first use
    kw = dict(embedding_size=128, epochs=1, walk_length=L, iterations=1, window_size=5,
              number_of_negative_samples=10, return_weight=rw, explore_weight=ew,
              learning_rate=0.01, random_state=42, verbose=False)
    fast = models.SkipGram(**kw)
    c, _, st = fast.fit_transform_device(g)
    # the block path: XCD cells at config 2's size, resident cells (LDS) at config 3's
    assert fast.last_plan is not None
    assert (fast.last_plan["slices"] > 8) == (nodes >= 100_000), fast.last_plan
    og = O.OracleGraph(host.row_ptr, host.col_idx)
    threads = min(16, len(os.sched_getaffinity(0)))
    rc, _, pairs = O.fit(og, O.WalkParams(L, 1, rw, ew, 100, 0),
                         O.TrainParams(0, 128, 128, 1, 10, 5, 0.01, 0.9, 6.0, 1, 128 ** -0.5), 42,
                         threads=threads)
    assert st["pairs"] == pairs
    res = _agreement(host, c.cpu().numpy()[:, :128], rc[:, :128])
    print(f"{shape}: default GPU fit vs oracle Hogwild ({threads} threads), {pairs} pairs: {res}")
    # measured (round 4): config 2's shape Spearman 0.96, mean |d cos| 0.078, AUROC 0.968 vs
    # 0.987 -- after ONE walk per node the 2 708-node fit is behind the CPU's: 8 x 2 000 row
    # updates in flight on 2 708 contextual rows lose part of what the rows outside the LDS
    # copies receive at the same moment (converged fits agree: 0.9950 vs 0.9961);
    # config 3's shape (resident cells): mean |d cos| 0.034
    assert res["spearman"] >= 0.9, res
    assert res["mean_abs"] <= 0.12, res
    assert abs(res["auc_got"] - res["auc_want"]) <= 0.04, res


@pytest.mark.timeout(900)
@pytest.mark.parametrize("cells,nodes,m,L,plan,rounds", [("xcd", 2_708, 2, 128, (1, 8), 1),
                                                         ("resident", 100_000, 5, 24, (2, 256), 4)])
def test_default_schedule_against_the_sequential_restatement_of_the_same_schedule(cells, nodes, m,
                                                                                  L, plan, rounds):
    """The GPU's parallel default and the oracle's sequential restatement of the block schedule
    train the SAME pairs with the SAME negatives (one round); they differ by what parallel
    execution does to the order of the updates -- and by what racing stores lose.  XCD cells at
    config 2's shape (1 x 8 cells of 338 rows; one wave per four rows, a row read again right
    before its stores); resident cells at the smallest size that gets them (100 k nodes: 2 x 256
    cells of 196 rows, each owned by one workgroup, under the rounds' placements; short walks,
    so that the sequential oracle finishes, but four rounds of them -- four walks a node, the
    cells reshuffled in between -- so that the cosines leave their random start and their rank
    correlation means something: gated)."""
    rw = ew = 1.0
    g = E.barabasi_albert(nodes, m, 42)
    host = g  # row_ptr / col_idx come to the host on first use
    d, w, k, lr = 128, 5, 10, 0.01
    wks = [ops.walks(g, ops.walk_params(L, 1, rw, ew), 42, 0, r * nodes, nodes)
           for r in range(rounds)]
    tables = {}
    for name in ("gpu", "oracle"):
        if name == "gpu":
            tp = ops.train_params(0, d, k, w, flags=1, ld=d)
            tr = BlockPartitionedTrainer(g, tp, d, d, 42, d ** -0.5, LoopbackComm(), "cuda:0",
                                         walk_length=L, window=w, parts=plan[0], slices=plan[1])
            for r, wk in enumerate(wks):
                tr.train_round(wk, 42, 0, lr, r * nodes)
        else:
            tp = O.TrainParams(0, d, d, 1, k, w, lr, 0.9, 6.0, 1, d ** -0.5)
            tr = BlockPartitionedTrainer(host, tp, d, d, 42, d ** -0.5, LoopbackComm(), "cpu",
                                         walk_length=L, window=w, backend=OracleBlockBackend(host),
                                         parts=plan[0], slices=plan[1])
            for r, wk in enumerate(wks):
                tr.train_round(wk.cpu(), 42, 0, lr, r * nodes)
        c, x = tr.gather_full()
        tables[name] = (c.cpu().numpy(), x.cpu().numpy())
    init_c = ops.init_table(nodes, d, 42, 0, d ** -0.5).cpu().numpy()
    init_x = ops.init_table(nodes, d, 42, 1, d ** -0.5).cpu().numpy()
    report = {}
    for t, init, label in ((0, init_c, "central"), (1, init_x, "contextual")):
        got, want = tables["gpu"][t], tables["oracle"][t]
        report[label] = dict(moved=float(np.linalg.norm(got - init) / np.linalg.norm(want - init)),
                             **_agreement(host, got, want, 50_000))
    print("parallel default vs sequential restatement of the block schedule:", report)
    # measured (round 4).  XCD cells at 2 708 nodes: central moved 0.98 x / Spearman 0.95,
    # contextual 0.82 x / 0.93 -- racing stores on a 338-row cell lose some updates (0.61 x with
    # a wave per row, 0.69 x before a row was read again right before its stores; an exact
    # accumulation in the same parallel order moves the contextual table 0.91 x: that is the
    # order, not a loss).  Resident cells, round 5 (four rounds under their placements, lockstep
    # phases): central 0.955 x / Spearman 0.971, contextual 0.954 x / 0.949, mean |d cos| 0.034 /
    # 0.043 (round 4's kernel moved the contextual table 0.85-0.87 x)
    assert 0.9 <= report["central"]["moved"] <= 1.1, report
    assert (0.9 if cells == "resident" else 0.75) <= report["contextual"]["moved"] <= 1.1, report
    for label, r in report.items():
        assert r["spearman"] >= 0.9, report
        assert r["mean_abs"] <= 0.12, report


def _central_store_losses(pairs_per_centre_and_cell, n_centres=60_000, d=128):
    """Every centre once (or twice in a row) in each of the 8 cells of a part, every context row
    unique, k = 0, a learning rate so small that the order does not matter: a centre's row is
    moved by the single-run store (row + gradient, write-through) of eight XCDs at eight moments
    of the launch.  Returns the central table's displacement relative to the sequential
    oracle's, per centre."""
    slices, record = 8, 32
    n_nodes = slices * n_centres * pairs_per_centre_and_cell
    g = E.barabasi_albert(n_nodes, 1, 3)
    host = g  # row_ptr / col_idx come to the host on first use
    og = O.OracleGraph(host.row_ptr, host.col_idx)
    plan = ops.block_plan(g, 1, 0, 1, slices, 8, 2, 1, record)
    oplan = O.block_plan(n_nodes, 1, 0, 1, slices, 8, 2, 1, record)
    rng = np.random.RandomState(5)
    words_l, offsets = [], [0]
    per_cell = n_centres * pairs_per_centre_and_cell
    for cell in range(slices):
        ctx = cell + slices * rng.permutation(n_nodes // slices)[:per_cell]   # unique rows
        centres = np.repeat(np.arange(n_centres), pairs_per_centre_and_cell)
        words_l.append(O.block_pack(np.full(per_cell, cell), centres, ctx, oplan))
        offsets.append(offsets[-1] + per_cell)
    words_h = np.concatenate(words_l).astype(np.uint64)
    off_h = np.asarray(offsets, dtype=np.uint64)
    pairs = torch.from_numpy(words_h.view(np.int64)).cuda()
    offs = torch.from_numpy(off_h.astype(np.int64)).cuda()
    lr = 1e-8
    tp = ops.train_params(0, d, 0, 2, flags=0, ld=d)
    otp = O.TrainParams(0, d, d, 1, 0, 2, 0.01, 0.9, 6.0, 0, d ** -0.5)
    c = torch.zeros((n_nodes, d), dtype=torch.float32, device="cuda")
    x = ops.init_table(n_nodes, d, 5, 1, 0.5).abs_()  # all positive: the gradients add up
    c_h, x_h = c.cpu().numpy().copy(), x.cpu().numpy().copy()
    ops.block_step(g, tp, plan, pairs, offs, None, None, c, x, 0, 0, 5, 0, lr)
    O.block_step(og, otp, oplan, words_h, off_h, None, None, c_h, x_h, 0, 0, 5, 0, lr)
    torch.cuda.synchronize()
    got, want = c.cpu().numpy()[:n_centres].astype(np.float64), c_h[:n_centres].astype(np.float64)
    assert np.abs(want).sum(1).min() > 0
    return got.sum(1) / want.sum(1)


@pytest.mark.parametrize("run", [1, 2])
def test_the_single_run_store_on_central_rows_loses_next_to_nothing(run):
    """block_kernels.h: a centre whose pairs in a cell form ONE run gets row + gradient written
    with a store instead of 128 atomics; seven other XCDs reach the same row through their own
    cells at other moments of the launch (every cell starts its stride order at its own offset).
    Counted: with every centre present in all 8 cells the central table must keep >= 99.9 % of
    the sequential displacement, and no centre may lose more than one of its eight updates.
    run = 1: records of single pairs (trained pair per group); run = 2: runs of two (the
    run-major loop's store)."""
    ratio = _central_store_losses(run)
    kept = float(ratio.mean())
    print(f"single-run store, runs of {run}: kept {kept:.6f} of the sequential displacement; "
          f"centres short of an update: {int((ratio < 0.95).sum())} of {len(ratio)}")
    assert kept >= 0.999, kept
    assert ratio.min() > 0.7, float(ratio.min())

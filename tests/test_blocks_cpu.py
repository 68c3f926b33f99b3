"""Block-partitioned multi-GPU trainer (embiggen_amd/distributed.py) on CPU: the oracle's
restatement of the schedule (oracle/gn2v_oracle.c, "block-partitioned SkipGram") as compute
stand-in (tests only), the host logic -- walk all-gather, part ownership, half-partition rotation,
final gather -- on threads and on torch.distributed (2 and 3 gloo ranks)."""
import os
import sys

import ctypes

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import embiggen_amd as E
from embiggen_amd.distributed import (BlockPartitionedTrainer, LoopbackComm, TorchComm,
                                      stripe_rows)
from oracle import oracle as O
from sharded_helpers import OracleBlockBackend, run_ranks

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
D, K, W, L = 8, 4, 3, 14
C_u64 = ctypes.c_uint64


def _graph(nodes=34):
    if nodes == 34:
        return E.karate_club()
    s, d = O.ba_edges(nodes, 3, 9)
    return E.CSRGraph.from_edge_list(s, d, number_of_nodes=nodes)


def _otp(flags=1):
    return O.TrainParams(0, D, D, 1, K, W, 0.02, 0.9, 6.0, flags, D ** -0.5)


def _train(comm, rounds=2, walks_per_round=9, nodes=34, slices=1, parts=None, record=4,
           stripes=1, group_parts=None, root=None):
    g = _graph(nodes)
    og = O.OracleGraph(g.row_ptr, g.col_idx)
    tr = BlockPartitionedTrainer(g, _otp(), D, D, 42, D ** -0.5, comm, "cpu", walk_length=L,
                                 window=W, backend=OracleBlockBackend(g), slices=slices,
                                 parts=parts, record=record, stripes=stripes,
                                 group_parts=group_parts)
    wp = O.WalkParams(L, 1, 0.25, 4.0, 100, 0)
    trained = 0
    for r in range(rounds):
        first = r * comm.world * walks_per_round
        mine = O.walks(og, wp, 42, 0, first + comm.rank * walks_per_round, walks_per_round)
        tr.train_round(torch.from_numpy(mine.view(np.int32)), 42, 0, 0.02, first)
        trained += tr.last_round["pairs_trained"]
    tables = [None if t is None else t.numpy().copy() for t in tr.gather_full(root=root)]
    return tables, trained, sorted(tr.held)


# ------------------------------------------------------------------ the restated schedule itself
@pytest.mark.parametrize("world,parts,slices", [(1, 1, 1), (2, 4, 1), (3, 6, 2), (4, 8, 8)])
def test_extraction_partitions_the_pairs_of_the_walks(world, parts, slices):
    g = _graph(97)
    og = O.OracleGraph(g.row_ptr, g.col_idx)
    walks = O.walks(og, O.WalkParams(L, 2, 0.5, 2.0, 100, 0), 3, 0, 0, 40)
    walks[::7, 9:] = O.SENTINEL  # walks that ended early
    want = O.walk_pairs(walks, W)
    got = []
    for rank in range(world):
        plan = O.block_plan(97, world, rank, parts, slices, L, W, 1, 4)
        words, offsets = O.block_extract(og, plan, walks, 3, 0, 0)
        raw, _ = O.block_extract(og, plan, walks, 3, 0, 0, sort=False)
        shift = O.block_sort_shift(plan)  # (cell, centre); resident plans: the centre's top bits
        assert shift == plan.ctx_bits + (max(0, plan.row_bits - 8) if slices > 16 else 0)
        keys = words >> np.uint64(shift)
        assert (np.diff(keys.astype(np.int64)) >= 0).all()
        # stable: pairs with equal keys keep the extraction (walk / position / slot) order
        order = np.argsort(raw >> np.uint64(shift), kind="stable")
        assert np.array_equal(words, raw[order])
        assert plan.key_bits <= 64 and not (words >> np.uint64(plan.key_bits)).any()
        cell, crow, vals, hot = O.block_unpack(words, plan)
        assert not hot.any() and np.array_equal(O.block_pack(cell, crow, vals, plan), words)
        assert offsets[-1] == len(words)
        for c in range(parts * slices):
            assert (cell[int(offsets[c]):int(offsets[c + 1])] == c).all()
        part, slc = cell // slices, cell % slices
        assert (vals % slices == slc).all()
        # a group of parts (cyclic) holds exactly the pairs of those parts, in the same order
        lo, n_grp = parts - 1, min(2, parts)
        grp, grp_off = O.block_extract(og, plan, walks, 3, 0, 0, part_lo=lo, part_n=n_grp)
        in_grp = ((part + parts - lo) % parts) < n_grp
        assert np.array_equal(grp, words[in_grp]) and grp_off[-1] == in_grp.sum()
        centre = crow.astype(np.int64) * world + rank
        context = vals.astype(np.int64) * parts + part
        got.append(np.stack([centre, context], 1))
    got = np.concatenate(got)
    as_sorted = lambda p: np.sort(p[:, 0].astype(np.int64) * 1000 + p[:, 1])  # noqa: E731
    assert np.array_equal(as_sorted(got), as_sorted(want))


def _alias_probabilities(table, n):
    """Exact law of a draw from one cell's alias table (thresholds on a 2^32 scale)."""
    thresh = (table & np.uint64(0xFFFFFFFE)).astype(np.float64)  # bit 0 is the hot-row flag
    alias = ((table >> np.uint64(32)) & np.uint64(0x7FFFFFFF)).astype(np.int64)
    keep = np.where(thresh >= 0xFFFFFFFE, 1.0, thresh / 2.0 ** 32)
    p = keep / n
    np.add.at(p, alias, (1.0 - keep) / n)
    return p


def test_alias_tables_are_degree_proportional_inside_a_cell():
    g = _graph(97)
    og = O.OracleGraph(g.row_ptr, g.col_idx)
    table, cell_rows, hub_bits, hot_list, hot_slot = O.block_alias(og, 4, 2, hot_rows=3)
    plain = O.block_alias(og, 4, 2)  # default: no hot rows, same law
    assert (plain[3] == 0xFFFFFFFF).all() and (plain[4] == 0xFF).all()
    assert not plain[2].any() and not (plain[0] & np.uint64(1)).any() and not (plain[0] >> np.uint64(63)).any()
    assert np.array_equal(plain[0] >> np.uint64(1) << np.uint64(1) & np.uint64(0x7FFFFFFFFFFFFFFF),
                          table >> np.uint64(1) << np.uint64(1) & np.uint64(0x7FFFFFFFFFFFFFFF))
    assert cell_rows[-1] == 97 and cell_rows[0] == 0
    indeg = np.bincount(og.col_idx, minlength=97).astype(np.float64)
    for cell in range(8):
        part, slc = cell // 2, cell % 2
        lo, hi = int(cell_rows[cell]), int(cell_rows[cell + 1])
        nodes = (slc + 2 * np.arange(hi - lo)) * 4 + part   # row i of the cell -> node id
        assert nodes.max() < 97 and hi - lo == len([v for v in range(97) if v % 4 == part
                                                      and (v // 4) % 2 == slc])
        p = _alias_probabilities(table[lo:hi], hi - lo)
        want = indeg[nodes] / indeg[nodes].sum()
        assert np.abs(p - want).max() < 1e-8 and abs(p.sum() - 1) < 1e-9
        alias_row = (table[lo:hi] >> np.uint64(32)) & np.uint64(0x7FFFFFFF)
        assert (alias_row < hi - lo).all()
        # hot rows: the three highest in-degrees of the cell (ties: the lower row first), in
        # slot order; every other slot of the list is empty
        order = sorted(range(hi - lo), key=lambda i: (-indeg[nodes[i]], i))[:3]
        assert list(hot_list[cell][:3]) == order and (hot_list[cell][3:] == 0xFFFFFFFF).all()
        hot = np.zeros(hi - lo, dtype=bool)
        hot[order] = True
        want_slot = np.full(hi - lo, 0xFF, dtype=np.uint8)
        want_slot[order] = np.arange(3)
        assert np.array_equal(hot_slot[lo:hi], want_slot)
        assert np.array_equal((table[lo:hi] & np.uint64(1)).astype(bool), hot)
        assert np.array_equal((table[lo:hi] >> np.uint64(63)).astype(bool), hot[alias_row.astype(int)])
        assert np.array_equal(((hub_bits[nodes >> 5] >> (nodes & 31)) & 1).astype(bool), hot)
    # empirical check of the sampler itself on one cell: chi-square against the degrees
    from scipy import stats
    lo, hi = int(cell_rows[3]), int(cell_rows[4])
    n = hi - lo
    r = np.array([O.lib().o_draw(C_u64(12345), C_u64(t)) for t in range(40000)], dtype=np.uint64)
    local = ((r >> np.uint64(32)) * np.uint64(n) >> np.uint64(32)).astype(np.int64)  # ~ mulhi
    e = table[lo:hi][local]
    take_alias = (r & np.uint64(0xFFFFFFFF)) >= (e & np.uint64(0xFFFFFFFE))
    local = np.where(take_alias, ((e >> np.uint64(32)) & np.uint64(0x7FFFFFFF)).astype(np.int64),
                     local)
    nodes = (1 + 2 * np.arange(n)) * 4 + 1
    want = indeg[nodes] / indeg[nodes].sum()
    counts = np.bincount(local, minlength=n).astype(np.float64)
    sel = want * 40000 >= 5
    assert stats.chisquare(counts[sel], want[sel] / want[sel].sum() * counts[sel].sum()).pvalue > 1e-4


def test_a_negative_on_the_context_or_centre_is_drawn_again():
    """o_block_step's negatives (oracle/gn2v_oracle.c block_negative; the device kernels follow
    it draw for draw): a cell-local negative that equals the pair's context or centre is drawn
    again, so (i) no negative ever IS the context or the centre, (ii) every pair of a cell with
    three rows or more trains its k negatives (the reference's graph-wide draw,
    node2vec_skipgram.py:101-102, loses ~degree / edges of them: nothing), (iii) given the
    context x and centre c the law is in-degree restricted to the cell's other rows --
    chi-square on the heaviest context of a cell, where the old skip rule dropped the largest
    share -- and (iv) cells of two rows (context + one mate) give up after O_NEG_ATTEMPTS draws
    only when that mate is the centre."""
    from scipy import stats

    g = _graph(97)
    og = O.OracleGraph(g.row_ptr, g.col_idx)
    indeg = np.bincount(og.col_idx, minlength=97).astype(np.float64)
    walks = O.walks(og, O.WalkParams(L, 1, 1.0, 1.0, 100, 0), 5, 0, 0, 97 * 30)
    parts, slices, k = 2, 2, 16
    plan = O.block_plan(97, 1, 0, parts, slices, L, W, 1, 4)
    words, offsets = O.block_extract(og, plan, walks, 5, 0, 0)
    alias, cell_rows = O.block_alias(og, parts, slices)[:2]
    tp = O.TrainParams(0, D, D, 1, k, W, 0.02, 0.9, 6.0, 1, D ** -0.5)
    cell, crow, vals, _ = O.block_unpack(words, plan)
    for part in range(parts):
        neg = O.block_negatives(og, tp, plan, words, offsets, alias, cell_rows, 0, part, 5, 0)
        lo, hi = int(offsets[part * slices]), int(offsets[(part + 1) * slices])
        ctx = vals[lo:hi].astype(np.int64) * parts + part
        centre = crow[lo:hi].astype(np.int64)
        # (ii) a sample is given up with probability (share of context + centre) ^ 8 only
        gone = neg == 0xFFFFFFFF
        assert neg.shape == (hi - lo, k) and gone.mean() < 2e-4, gone.mean()
        assert (neg != ctx[:, None]).all() and (neg != centre[:, None]).all()  # (i)
        slc = (neg.astype(np.int64) // parts) % slices
        assert ((neg % parts == part) | gone).all()
        assert ((slc == (ctx[:, None] // parts) % slices) | gone).all()
        # (iii) the context that carries the most pairs of the part, centres outside its cell
        x = np.bincount(ctx).argmax()
        in_cell = lambda v: (v % parts == part) & ((v // parts) % slices == (x // parts) % slices)  # noqa: E731
        sel = (ctx == x) & ~in_cell(centre)
        mates = np.array([v for v in range(97) if in_cell(np.int64(v)) and v != x])
        want = indeg[mates] / indeg[mates].sum()
        drawn = neg[sel].ravel()
        counts = np.bincount(drawn[drawn != 0xFFFFFFFF], minlength=97)[mates].astype(np.float64)
        keep = want * counts.sum() >= 5
        assert counts.sum() >= 0.999 * sel.sum() * k and counts[~keep].sum() <= 0.02 * counts.sum()
        pv = stats.chisquare(counts[keep], want[keep] / want[keep].sum() * counts[keep].sum()).pvalue
        assert pv > 1e-4, (part, x, pv)
        # ... under the old rule this context would have lost its own share of the cell's draws
        assert indeg[x] / (indeg[x] + indeg[mates].sum()) > 0.05
    # (iv) two-row cells: 4 nodes, 1 part x 2 slices; path 0 - 1 - 2 - 3 (cells {0, 2}, {1, 3})
    g4 = E.CSRGraph.from_edge_list(np.array([0, 1, 2]), np.array([1, 2, 3]), number_of_nodes=4)
    og4 = O.OracleGraph(g4.row_ptr, g4.col_idx)
    plan4 = O.block_plan(4, 1, 0, 1, 2, 2, 1, 1, 4)
    w4 = np.array([[0, 2], [1, 2], [0, 1]], dtype=np.uint32)  # (centre 0, ctx 2): the mate IS the centre
    words4, off4 = O.block_extract(og4, plan4, w4, 5, 0, 0)
    alias4, rows4 = O.block_alias(og4, 1, 2)[:2]
    tp4 = O.TrainParams(0, D, D, 1, 3, 1, 0.02, 0.9, 6.0, 1, D ** -0.5)
    neg4 = O.block_negatives(og4, tp4, plan4, words4, off4, alias4, rows4, 0, 0, 5, 0)
    _, crow4, vals4, _ = O.block_unpack(words4, plan4)
    for (c, x), row in zip(zip(crow4.astype(int), vals4.astype(int)), neg4):
        mate = {0: 2, 2: 0, 1: 3, 3: 1}[x]
        # (the context of in-degree 2 beside a mate of in-degree 1 is drawn 8 times in a row
        # with probability (2 / 3) ^ 8: such a sample is given up as well)
        assert ((row == 0xFFFFFFFF) if mate == c else np.isin(row, (mate, 0xFFFFFFFF))).all(), \
            (c, x, row)
    assert (neg4 != 0xFFFFFFFF).sum() >= neg4.size // 2


def test_record_visiting_order_is_a_permutation():
    for R in (1, 2, 3, 10, 97, 1000, 4096, 65537):
        A = O.block_record_stride(R)
        assert sorted((t * A) % R for t in range(R)) == list(range(R))
        if R > 10:  # consecutive tickets land far apart
            assert min(A, R - A) > R // 4


def test_step_properties_zero_lr_counts_and_untouched_rows():
    g = _graph(97)
    og = O.OracleGraph(g.row_ptr, g.col_idx)
    walks = O.walks(og, O.WalkParams(L, 1, 1.0, 1.0, 100, 0), 5, 0, 0, 30)
    plan = O.block_plan(97, 1, 0, 2, 1, L, W, 1, 4)
    words, offsets = O.block_extract(og, plan, walks, 5, 0, 0)
    alias, cell_rows = O.block_alias(og, 2, 1)[:2]
    c = O.init_table_rows(97, D, D, 5, 0, 0.3, 0, 1)
    assert np.array_equal(c, O.init_table(97, D, D, 5, 0, 0.3))
    parts = [O.init_table_rows(stripe_rows(97, p, 2), D, D, 5, 1, 0.3, p, 2) for p in range(2)]
    full = O.init_table(97, D, D, 5, 1, 0.3)
    assert np.array_equal(parts[0], full[0::2]) and np.array_equal(parts[1], full[1::2])
    c0, x0 = c.copy(), [p.copy() for p in parts]
    n0 = O.block_step(og, _otp(), plan, words, offsets, alias, cell_rows, c, parts[0], 0, 0,
                      5, 0, 0.0)
    assert np.array_equal(c, c0) and np.array_equal(parts[0], x0[0])  # lr = 0: identity
    n1 = sum(O.block_step(og, _otp(), plan, words, offsets, alias, cell_rows, c, parts[p],
                          0, p, 5, 0, 0.05) for p in range(2))
    assert n0 == offsets[1] and n1 == len(words) == len(O.walk_pairs(walks, W))
    _, crow, vals, _ = O.block_unpack(words, plan)
    centres = np.unique(crow).astype(np.int64)
    untouched = np.setdiff1d(np.arange(97), centres)
    assert np.array_equal(c[untouched], c0[untouched]) and not np.array_equal(c, c0)
    indeg = np.bincount(og.col_idx, minlength=97)
    for p in range(2):  # contextual rows move only for contexts and nodes an edge points to
        seg = slice(int(offsets[p]), int(offsets[p + 1]))
        may = np.union1d(vals[seg], np.nonzero(indeg[p::2])[0])
        rest = np.setdiff1d(np.arange(len(parts[p])), may)
        assert np.array_equal(parts[p][rest], x0[p][rest])


# ------------------------------------------------------------------ trainer host logic
def test_world_one_trainer_is_the_plain_sequence_of_block_steps():
    (c, x), trained, held = _train(LoopbackComm(), parts=2)
    g = _graph()
    og = O.OracleGraph(g.row_ptr, g.col_idx)
    plan = O.block_plan(34, 1, 0, 2, 1, L, W, 1, 4)
    alias, cell_rows = O.block_alias(og, 2, 1)[:2]
    rc = O.init_table(34, D, D, 42, 0, D ** -0.5)
    rx = O.init_table(34, D, D, 42, 1, D ** -0.5)
    parts = [np.ascontiguousarray(rx[p::2]) for p in range(2)]
    wp = O.WalkParams(L, 1, 0.25, 4.0, 100, 0)
    total = 0
    for r in range(2):
        walks = O.walks(og, wp, 42, 0, r * 9, 9)
        words, offsets = O.block_extract(og, plan, walks, 42, 0, r * 9)
        for p in range(2):
            total += O.block_step(og, _otp(), plan, words, offsets, alias, cell_rows, rc,
                                  parts[p], r, p, 42, 0, 0.02)
    rx[0::2], rx[1::2] = parts
    assert np.array_equal(c, rc) and np.array_equal(x, rx)
    assert trained == total == 2 * 9 * (2 * W * L - W * (W + 1)) and held == [0, 1]


@pytest.mark.parametrize("stripes,parts,slices", [(2, 2, 1), (3, 2, 2), (8, 1, 1)])
def test_centre_stripes_are_the_sequence_of_block_steps_of_that_many_ranks(stripes, parts, slices):
    """One GPU, `stripes` centre stripes trained one after the other (BlockPartitionedTrainer
    stripes=): per round, for every stripe j, the pairs whose centre is j mod stripes are
    extracted from ALL the round's walks with the plan of rank j of a world of `stripes`, and
    trained part by part on the stripe's rows of the one central table; RNG stream
    round * stripes + j, the rotation of the parts simply continuing."""
    (c, x), trained, held = _train(LoopbackComm(), nodes=97, parts=parts, slices=slices,
                                   stripes=stripes)
    g = _graph(97)
    og = O.OracleGraph(g.row_ptr, g.col_idx)
    alias, cell_rows = O.block_alias(og, parts, slices)[:2]
    rc = O.init_table(97, D, D, 42, 0, D ** -0.5)
    rx = O.init_table(97, D, D, 42, 1, D ** -0.5)
    ctx = [np.ascontiguousarray(rx[p::parts]) for p in range(parts)]
    wp = O.WalkParams(L, 1, 0.25, 4.0, 100, 0)
    total, episode = 0, 0
    for r in range(2):
        walks = O.walks(og, wp, 42, 0, r * 9, 9)
        for j in range(stripes):
            plan = O.block_plan(97, stripes, j, parts, slices, L, W, 1, 4)
            words, offsets = O.block_extract(og, plan, walks, 42, 0, r * 9)
            mine = np.ascontiguousarray(rc[j::stripes])
            for _ in range(parts):
                p = episode % parts
                total += O.block_step(og, _otp(), plan, words, offsets, alias, cell_rows,
                                      mine, ctx[p], r * stripes + j, p, 42, 0, 0.02)
                episode += 1
            rc[j::stripes] = mine
    for p in range(parts):
        rx[p::parts] = ctx[p]
    assert np.array_equal(c, rc) and np.array_equal(x, rx)
    # last_round counts the pairs of a whole round (all stripes); two rounds were trained
    assert trained == total == 2 * 9 * (2 * W * L - W * (W + 1)) and held == list(range(parts))
    with pytest.raises(ValueError):
        run_ranks(2, lambda comm: _train(comm, nodes=97, stripes=2))


@pytest.mark.parametrize("world,nodes,per_rank", [(2, 34, 2), (3, 34, 2), (4, 97, 2), (2, 97, 4),
                                                  (3, 97, 3)])
def test_ranks_cover_every_pair_once_and_assemble_identical_tables(world, nodes, per_rank):
    parts = per_rank * world
    sims = run_ranks(world, lambda comm: _train(comm, rounds=1, nodes=nodes, parts=parts))
    assert sum(s[1] for s in sims) == world * 9 * (2 * W * L - W * (W + 1))
    for s in sims[1:]:
        assert np.array_equal(s[0][0], sims[0][0][0]) and np.array_equal(s[0][1], sims[0][0][1])
    # after `parts` episodes rank r holds parts P r - 1 ... P r + P - 2 (one hop from home)
    for r, s in enumerate(sims):
        assert s[2] == sorted((per_rank * r - 1 + j) % parts for j in range(per_rank))
    assert sorted(p for s in sims for p in s[2]) == list(range(parts))
    init = O.init_table(nodes, D, D, 42, 1, D ** -0.5)
    assert np.abs(sims[0][0][1] - init).max() > 1e-3
    assert np.isfinite(sims[0][0][0]).all() and np.isfinite(sims[0][0][1]).all()


def test_slices_change_the_negative_cells_not_the_bookkeeping():
    plain = run_ranks(2, lambda comm: _train(comm, rounds=1, nodes=97))
    sliced = run_ranks(2, lambda comm: _train(comm, rounds=1, nodes=97, slices=2))
    assert plain[0][1] + plain[1][1] == sliced[0][1] + sliced[1][1]
    assert not np.array_equal(plain[0][0][1], sliced[0][0][1])


@pytest.mark.parametrize("world,parts,group_parts", [(1, 4, 1), (1, 4, 3), (2, 4, 1), (2, 8, 3),
                                                     (3, 6, 4)])
def test_groups_of_parts_change_the_memory_not_the_result(world, parts, group_parts):
    """A round extracted, sorted and trained a group of parts at a time (each rank's groups
    start at its own first part and wrap round) is bit-equal to the round prepared at once: a
    cell's pairs keep their order, the negatives' streams are keyed by the position inside the
    cell."""
    whole = run_ranks(world, lambda comm: _train(comm, nodes=97, parts=parts, slices=2))
    grouped = run_ranks(world, lambda comm: _train(comm, nodes=97, parts=parts, slices=2,
                                                   group_parts=group_parts))
    for a, b in zip(whole, grouped):
        assert np.array_equal(a[0][0], b[0][0]) and np.array_equal(a[0][1], b[0][1])
        assert a[1] == b[1] and a[2] == b[2]


@pytest.mark.parametrize("world,root", [(2, 0), (3, 1)])
def test_gather_to_one_rank_only(world, root):
    """gather_full(root=r): rank r assembles the tables every rank would assemble; the others
    get nothing."""
    every = run_ranks(world, lambda comm: _train(comm, nodes=97))
    rooted = run_ranks(world, lambda comm: _train(comm, nodes=97, root=root))
    for r in range(world):
        if r == root:
            assert np.array_equal(rooted[r][0][0], every[r][0][0])
            assert np.array_equal(rooted[r][0][1], every[r][0][1])
        else:
            assert rooted[r][0] == [None, None]


def _gloo_worker(rank, world, port, out_dir, parts=None, root=None, group_parts=None):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    (c, x), _, _ = _train(TorchComm(), parts=parts, root=root, group_parts=group_parts)
    if c is not None:
        np.save(os.path.join(out_dir, f"c{rank}.npy"), c)
        np.save(os.path.join(out_dir, f"x{rank}.npy"), x)
    dist.barrier()
    dist.destroy_process_group()


def _free_port():
    import socket

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.parametrize("world,parts", [(2, None), (3, None), (2, 8)])
def test_gloo_ranks_equal_the_in_process_simulation(tmp_path, world, parts):
    """No row is ever shared between ranks, so the distributed run is exactly the simulation
    (world 3: ragged parts of 6 / 6 / 6 / 6 / 5 / 5 rows travel round the ring; two rounds, so
    the rotation continues across the round boundary; 2 ranks x 4 parts each: a part arrives
    three episodes before it is trained)."""
    mp.spawn(_gloo_worker, args=(world, _free_port(), str(tmp_path), parts), nprocs=world,
             join=True)
    sim = run_ranks(world, lambda comm: _train(comm, parts=parts))
    for r in range(world):
        assert np.array_equal(np.load(tmp_path / f"c{r}.npy"), sim[r][0][0])
        assert np.array_equal(np.load(tmp_path / f"x{r}.npy"), sim[r][0][1])
    init = O.init_table(34, D, D, 42, 0, D ** -0.5)
    assert np.abs(sim[0][0][0] - init).max() > 1e-3


def test_gloo_ranks_gather_to_one_rank_and_train_in_groups(tmp_path):
    """3 gloo ranks, 6 parts prepared two at a time, the result assembled on rank 1 only (central
    partitions and context parts sent point to point, one at a time): rank 1 holds exactly what
    the simulation computes, the other ranks write nothing."""
    mp.spawn(_gloo_worker, args=(3, _free_port(), str(tmp_path), 6, 1, 2), nprocs=3, join=True)
    sim = run_ranks(3, lambda comm: _train(comm, parts=6))
    assert sorted(os.listdir(tmp_path)) == ["c1.npy", "x1.npy"]
    assert np.array_equal(np.load(tmp_path / "c1.npy"), sim[1][0][0])
    assert np.array_equal(np.load(tmp_path / "x1.npy"), sim[1][0][1])


# ------------------------------------------------------------------ placement of a round (round 5)
@pytest.mark.parametrize("n,classes", [(34, 1), (97, 1), (97, 4), (203, 6), (1000, 16)])
def test_placement_is_a_seeded_permutation_inside_its_classes(n, classes):
    """o_block_placement: a bijection of the node ids that keeps every node in its residue class
    modulo `classes` (several ranks: a row never leaves its part), reproducible, different for
    every round and seed."""
    place, inv = O.block_placement(n, classes, 42, 3)
    assert sorted(place.tolist()) == list(range(n))
    assert np.array_equal(inv[place], np.arange(n, dtype=np.uint32))
    assert np.array_equal(place % classes, np.arange(n) % classes)
    again = O.block_placement(n, classes, 42, 3)
    assert np.array_equal(place, again[0]) and np.array_equal(inv, again[1])
    for seed, rnd in ((42, 4), (43, 3)):
        assert not np.array_equal(place, O.block_placement(n, classes, seed, rnd)[0])


def test_cell_mates_change_from_round_to_round():
    """What the placement is for: under the fixed striping a node shares its cell with the same
    nodes for a whole fit (x = y modulo parts x slices); under the placement of round r the
    cell-mates of a node are another random set every round -- over 64 rounds a node of a
    1 000-node graph in cells of ~20 rows meets most of the graph, every other node about
    equally often."""
    n, parts, slices = 1000, 2, 25
    cells = parts * slices
    met = np.zeros((n, n), dtype=np.int32)
    rounds = 64
    for r in range(rounds):
        place = O.block_placement(n, 1, 7, r)[0].astype(np.int64)
        cell = (place % parts) * slices + (place // parts) % slices
        met += (cell[:, None] == cell[None, :])
    np.fill_diagonal(met, 0)
    per_node = (met > 0).sum(1)
    # 64 rounds x 19 mates = 1 216 draws from 999 nodes: ~70 % of the graph met at least once
    assert per_node.min() > 0.6 * n and per_node.mean() > 0.68 * n
    # and nobody is a mate much more often than chance (64 * 19 / 999 = 1.2 rounds)
    assert met.max() <= 9
    # the fixed striping, for contrast: always the same 19 mates
    ids = np.arange(n)
    fixed = (ids % parts) * slices + (ids // parts) % slices
    assert ((fixed[:, None] == fixed[None, :]).sum(1) - 1).max() == n // cells - 1 + (n % cells > 0)
    # classes = parts (several ranks): a node only ever meets nodes of its own part
    place = O.block_placement(n, parts, 7, 5)[0].astype(np.int64)
    assert np.array_equal(place % parts, ids % parts)


@pytest.mark.parametrize("world,parts,slices,group_parts", [(1, 2, 24, None), (1, 3, 17, 2),
                                                            (2, 4, 24, None), (3, 6, 20, 4)])
def test_trainer_under_a_placement_equals_the_restated_schedule(world, parts, slices, group_parts):
    """Plans of more than 16 slices (resident cells) are trained under a placement per round:
    over the whole graph on one rank (the contextual table stays one table in node order),
    inside the classes modulo `parts` with several ranks (the parts travel as before).  The
    trainer -- threads as ranks -- against the schedule restated call by call: placement,
    alias tables and extraction of the round, then every (rank, part) block once."""
    nodes, rounds, wpr = 203, 2, 9
    g = _graph(nodes)
    og = O.OracleGraph(g.row_ptr, g.col_idx)
    sims = run_ranks(world, lambda comm: _train(comm, rounds=rounds, walks_per_round=wpr,
                                                nodes=nodes, parts=parts, slices=slices,
                                                group_parts=group_parts, record=8))
    for s in sims[1:]:
        assert np.array_equal(s[0][0], sims[0][0][0]) and np.array_equal(s[0][1], sims[0][0][1])
    assert sum(s[1] for s in sims) == rounds * world * wpr * (2 * W * L - W * (W + 1))
    # the restatement
    rc = O.init_table(nodes, D, D, 42, 0, D ** -0.5)
    rx = O.init_table(nodes, D, D, 42, 1, D ** -0.5)
    classes = parts if world > 1 else 1
    wp = O.WalkParams(L, 1, 0.25, 4.0, 100, 0)
    per_rank = parts // world
    for r in range(rounds):
        walks = O.walks(og, wp, 42, 0, r * world * wpr, world * wpr)
        place, inv = O.block_placement(nodes, classes, 42, r)
        alias, cell_rows = O.block_alias(og, parts, slices, 0, inv=inv)[:2]
        plans = [O.block_plan(nodes, world, rank, parts, slices, L, W, 1, 8)
                 for rank in range(world)]
        prepared = [O.block_extract(og, plans[rank], walks, 42, 0, r * world * wpr, place=place)
                    for rank in range(world)]
        mine = [np.ascontiguousarray(rc[rank::world]) for rank in range(world)]
        for e in range(parts):  # episode by episode: a part visits the ranks one after the other
            for rank in range(world):
                p = (per_rank * rank + e) % parts
                O.block_step(og, _otp(), plans[rank], prepared[rank][0], prepared[rank][1], alias,
                             cell_rows, mine[rank], rx, r * world + rank, p, 42, 0, 0.02, inv=inv,
                             natural=True)
        for rank in range(world):
            rc[rank::world] = mine[rank]
    assert np.array_equal(sims[0][0][0], rc) and np.array_equal(sims[0][0][1], rx)
    assert np.abs(rx - O.init_table(nodes, D, D, 42, 1, D ** -0.5)).max() > 1e-3


def test_gloo_ranks_under_a_placement_equal_the_in_process_simulation(tmp_path):
    """2 gloo ranks, 4 parts x 24 slices (resident cells: a placement per round inside the
    classes modulo 4, the alias tables rebuilt per round on every rank alike)."""
    mp.spawn(_gloo_worker_placed, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    sim = run_ranks(2, lambda comm: _train(comm, nodes=97, parts=4, slices=24, record=8))
    for r in range(2):
        assert np.array_equal(np.load(tmp_path / f"c{r}.npy"), sim[r][0][0])
        assert np.array_equal(np.load(tmp_path / f"x{r}.npy"), sim[r][0][1])


def _gloo_worker_placed(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    (c, x), _, _ = _train(TorchComm(), nodes=97, parts=4, slices=24, record=8)
    np.save(os.path.join(out_dir, f"c{rank}.npy"), c)
    np.save(os.path.join(out_dir, f"x{rank}.npy"), x)
    dist.barrier()
    dist.destroy_process_group()

"""GPU: `sgns_resident_kernel` -- the default training kernel of every graph from 100 k to 115 M
nodes (BASELINE configs 3, 4, 5a, 5; nine tenths of the bench's GPU time) -- against the oracle's
restatement of the block schedule (`oracle/gn2v_oracle.c` o_block_step), element by element:

* its deterministic instantiation (GN2V_TRAIN_DETERMINISTIC on a resident plan: one workgroup,
  cells in order, records in stride order, samples in order, rows in LDS) with k > 0 on real
  walks -- the run-major loop -- and on records of unique centres -- the pair-per-group loop with
  its side-by-side samples, packed dot products, v_rcp sigmoid, prefetched central rows and the
  transposed atomic hand-over --, d in {16, 100, 128, 256};
* its parallel form (default flags) on inputs whose result does not depend on the order:
  collision-free pairs, and pairs that share centres inside a step of four;
* and what the parallel form LOSES inside a cell, counted row by row (a row every pair of the cell
  hits; a cell whose top row receives 4 % of the samples): see test_in_cell_losses_*.

The call all of this replaces: embedders/ensmallen_embedders/node2vec.py:99."""
import numpy as np
import pytest
import torch

import embiggen_amd as E
from embiggen_amd import _lib, ops
from embiggen_amd.distributed import stripe_rows
from oracle import oracle as O

pytestmark = pytest.mark.gpu
DET = _lib.TRAIN_DETERMINISTIC


def _ba(nodes, m=3, seed=9):
    s, d = O.ba_edges(nodes, m, seed)
    return E.CSRGraph.from_edge_list(s, d, number_of_nodes=nodes)


def _words(t):
    return t.cpu().numpy().view(np.uint64)


def _dev_words(words):
    return torch.from_numpy(np.ascontiguousarray(words).view(np.int64)).cuda()


def _resident_launches(g):
    return ops.stats_read(g)["resident_launches"]


def _step_real_walks(g, og, d, k, parts, slices, record, flags, lr=0.05, n_walks=400, wl=20,
                     window=4, scale_free=True):
    """One round of real walks through gn2v_block_step (part by part) and through the oracle."""
    n = g.get_number_of_nodes()
    ld = (d + 3) // 4 * 4 if d <= 16 else (d + 31) // 32 * 32
    wk = ops.walks(g, ops.walk_params(wl, 2, 0.5, 2.0), 11, 0, 0, n_walks)
    plan = ops.block_plan(g, 1, 0, parts, slices, wl, window, 1, record)
    oplan = O.block_plan(n, 1, 0, parts, slices, wl, window, 1, record)
    work, offsets = ops.block_count(g, plan, wk, 11, 0, 0)
    alias, cell_rows, hub_bits, hot_list, hot_slot = ops.block_alias(g, plan)
    pairs = ops.block_extract(g, plan, wk, 11, 0, 0, work, int(offsets[-1]), hub_bits=hub_bits)
    sf = 1 if scale_free else 0
    tp = ops.train_params(0, d, k, window, flags=sf | flags, ld=ld)
    otp = O.TrainParams(0, d, ld, 1, k, window, 0.01, 0.9, 6.0, sf, d ** -0.5)
    c = ops.init_table(n, d, 11, 0, d ** -0.5, ld=ld)
    c0 = c.cpu().numpy().copy()
    c_h = c0.copy()
    rw, ro = _words(pairs), offsets.cpu().numpy().astype(np.uint64)
    rp, rpo = alias.cpu().numpy().view(np.uint64), cell_rows.cpu().numpy().astype(np.uint64)
    got_x, ref_x = [], []
    ops.stats_reset(g)
    trained = 0
    for part in range(parts):
        x = ops.init_table_rows(stripe_rows(n, part, parts), d, 11, 1, d ** -0.5, part, parts, ld=ld)
        x_h = x.cpu().numpy().copy()
        ops.block_step(g, tp, plan, pairs, offsets, alias, cell_rows, c, x, 3, part, 11, 0, lr)
        trained += O.block_step(og, otp, oplan, rw, ro, rp, rpo, c_h, x_h, 3, part, 11, 0, lr)
        got_x.append(x.cpu().numpy())
        ref_x.append(x_h)
    torch.cuda.synchronize()
    st = ops.stats_read(g)
    assert st["pairs"] == trained and trained > 0
    assert st["resident_launches"] == parts, st  # the kernel under test is the one that ran
    assert np.abs(c_h - c0).max() > 1e-3
    return c.cpu().numpy(), got_x, c_h, ref_x


@pytest.mark.parametrize("d,k,parts,slices,record", [
    (16, 4, 2, 32, 16), (100, 5, 2, 32, 32), (128, 10, 2, 64, 32), (128, 3, 3, 17, 8),
    (256, 5, 2, 32, 16), (64, 0, 1, 40, 32), (320, 4, 2, 40, 16), (512, 5, 2, 32, 32),
    (32, 70, 2, 32, 8), (128, 64, 1, 40, 8)])
def test_deterministic_resident_step_matches_oracle(d, k, parts, slices, record):
    """Real walks (runs of equal centre, k > 0 negatives from the cell's alias table, negatives
    that fall on the context or the centre and are drawn again, cells of 8-30 rows in LDS):
    <= 1e-5 per element.  k >= 64: sample lists longer than a wave (the null list -- what a
    group without a pair scores -- has more than 64 entries)."""
    g = _ba(1999)
    og = O.OracleGraph(g.row_ptr, g.col_idx)
    c, xs, c_h, xs_h = _step_real_walks(g, og, d, k, parts, slices, record, DET)
    # (k >= 64: 70 negatives a pair at this learning rate grow the rows to |2|: the bound is
    # relative to the largest element)
    tol = 1e-5 * max(1.0, float(np.abs(c_h).max()), max(float(np.abs(x).max()) for x in xs_h))
    assert np.abs(c - c_h).max() < tol
    for x, x_h in zip(xs, xs_h):
        assert np.abs(x - x_h).max() < tol


def test_deterministic_resident_step_with_uniform_negatives_and_degree_normalised_rate():
    g = _ba(1999)
    og = O.OracleGraph(g.row_ptr, g.col_idx)
    n, d, k = 1999, 32, 4
    wk = ops.walks(g, ops.walk_params(20, 2, 0.5, 2.0), 11, 0, 0, 300)
    plan = ops.block_plan(g, 1, 0, 2, 32, 20, 4, 1, 16)
    oplan = O.block_plan(n, 1, 0, 2, 32, 20, 4, 1, 16)
    work, offsets = ops.block_count(g, plan, wk, 11, 0, 0)
    pairs = ops.block_extract(g, plan, wk, 11, 0, 0, work, int(offsets[-1]))
    tp = ops.train_params(0, d, k, 4, flags=DET | _lib.TRAIN_NORM_LR, ld=d)
    otp = O.TrainParams(0, d, d, 1, k, 4, 0.01, 0.9, 6.0, O.FLAG_NORM_LR, d ** -0.5)
    c = ops.init_table(n, d, 11, 0, d ** -0.5)
    c_h = c.cpu().numpy().copy()
    rw, ro = _words(pairs), offsets.cpu().numpy().astype(np.uint64)
    ops.stats_reset(g)
    for part in range(2):
        x = ops.init_table_rows(stripe_rows(n, part, 2), d, 11, 1, d ** -0.5, part, 2)
        x_h = x.cpu().numpy().copy()
        ops.block_step(g, tp, plan, pairs, offsets, None, None, c, x, 3, part, 11, 0, 0.5)
        O.block_step(og, otp, oplan, rw, ro, None, None, c_h, x_h, 3, part, 11, 0, 0.5)
        torch.cuda.synchronize()
        assert np.abs(x.cpu().numpy() - x_h).max() < 1e-5
    assert _resident_launches(g) == 2
    assert np.abs(c.cpu().numpy() - c_h).max() < 1e-5


def _unique_centre_pairs(n_nodes, parts, slices, oplan, per_cell, seed=3):
    """Sorted pair words with unique, ascending centre rows -- every record then consists of
    single-pair runs -- and contexts drawn WITH repetition from the cell's rows."""
    rng = np.random.RandomState(seed)
    words_l, offsets, centre = [], [0], 0
    for cell in range(parts * slices):
        part, slc = divmod(cell, slices)
        rows = stripe_rows(stripe_rows(n_nodes, part, parts), slc, slices)
        ctx = slc + slices * rng.randint(0, rows, per_cell)  # row inside the part
        words_l.append(O.block_pack(np.full(per_cell, cell), centre + np.arange(per_cell), ctx,
                                    oplan))
        centre += per_cell
        offsets.append(offsets[-1] + per_cell)
    assert centre <= n_nodes
    return np.concatenate(words_l).astype(np.uint64), np.asarray(offsets, dtype=np.uint64)


@pytest.mark.parametrize("d,k", [(16, 4), (100, 5), (128, 10), (128, 1), (256, 6), (400, 5),
                                 (512, 10), (512, 2), (64, 66), (128, 65)])
def test_deterministic_pair_per_group_loop_matches_oracle(d, k):
    """Records of unique centres take the pair-per-group loop -- in the deterministic
    instantiation with the four groups taking turns: score_sample_pair (two samples side by
    side, packed dot products, v_rcp sigmoid), the single sample left over when 1 + k is odd, the
    one-row-twice branch (contexts and negatives repeat inside ~20-row cells), the prefetched
    central rows, the transposed atomic hand-over: <= 1e-5 against the oracle with k > 0."""
    n, parts, slices, record = 1999, 2, 48, 16
    g = _ba(n)
    og = O.OracleGraph(g.row_ptr, g.col_idx)
    ld = (d + 31) // 32 * 32 if d > 16 else d
    plan = ops.block_plan(g, 1, 0, parts, slices, 8, 2, 1, record)
    oplan = O.block_plan(n, 1, 0, parts, slices, 8, 2, 1, record)
    alias, cell_rows, _, _, _ = ops.block_alias(g, plan)
    words_h, off_h = _unique_centre_pairs(n, parts, slices, oplan, per_cell=19)
    pairs, offs = _dev_words(words_h), torch.from_numpy(off_h.astype(np.int64)).cuda()
    tp = ops.train_params(0, d, k, 2, flags=1 | DET, ld=ld)
    otp = O.TrainParams(0, d, ld, 1, k, 2, 0.01, 0.9, 6.0, 1, d ** -0.5)
    c = ops.init_table(n, d, 5, 0, 0.5, ld=ld)
    c_h = c.cpu().numpy().copy()
    rp, rpo = alias.cpu().numpy().view(np.uint64), cell_rows.cpu().numpy().astype(np.uint64)
    ops.stats_reset(g)
    for part in range(parts):
        x = ops.init_table_rows(stripe_rows(n, part, parts), d, 5, 1, 0.5, part, parts, ld=ld)
        x_h = x.cpu().numpy().copy()
        x0 = x_h.copy()
        ops.block_step(g, tp, plan, pairs, offs, alias, cell_rows, c, x, 2, part, 5, 0, 0.1)
        O.block_step(og, otp, oplan, words_h, off_h, rp, rpo, c_h, x_h, 2, part, 5, 0, 0.1)
        torch.cuda.synchronize()
        assert np.abs(x.cpu().numpy() - x_h).max() < 1e-5
        assert np.abs(x_h - x0).max() > 1e-3
    st = ops.stats_read(g)
    # one central hand-over per pair: the pair-per-group loop, not the run-major one
    assert st["resident_launches"] == parts and st["pairs"] == st["centres"] == len(words_h)
    assert np.abs(c.cpu().numpy() - c_h).max() < 1e-5


@pytest.mark.parametrize("d,per_cell,slices", [(8, 150, 512), (128, 150, 512), (100, 120, 512),
                                               (256, 70, 1024), (512, 36, 2048), (330, 36, 2048)])
def test_parallel_resident_step_on_collision_free_pairs(d, per_cell, slices):
    """The kernel as it ships (default flags, sixteen waves per cell, LDS cursor, stride order
    with a start offset, pair per group, central rows by atomics) on 1 024 cells of <= 157 rows
    with every centre and every context row used once, k = 0: nothing can collide, so it must
    equal the sequential oracle -- loads into LDS, write-back, tickets, hand-over and all."""
    parts, record = 2, 16
    n = 160_005
    g = _ba(n)
    og = O.OracleGraph(g.row_ptr, g.col_idx)
    ld = (d + 31) // 32 * 32 if d > 16 else d
    plan = ops.block_plan(g, 1, 0, parts, slices, 8, 2, 1, record)
    oplan = O.block_plan(n, 1, 0, parts, slices, 8, 2, 1, record)
    rng = np.random.RandomState(3)
    words_l, offsets, centre = [], [0], 0
    for cell in range(parts * slices):
        part, slc = divmod(cell, slices)
        rows = stripe_rows(stripe_rows(n, part, parts), slc, slices)
        assert rows >= per_cell
        ctx = slc + slices * rng.permutation(rows)[:per_cell]
        words_l.append(O.block_pack(np.full(per_cell, cell), centre + np.arange(per_cell), ctx, oplan))
        centre += per_cell
        offsets.append(offsets[-1] + per_cell)
    words_h = np.concatenate(words_l).astype(np.uint64)
    off_h = np.asarray(offsets, dtype=np.uint64)
    pairs, offs = _dev_words(words_h), torch.from_numpy(off_h.astype(np.int64)).cuda()
    tp = ops.train_params(0, d, 0, 2, flags=0, ld=ld)
    otp = O.TrainParams(0, d, ld, 1, 0, 2, 0.01, 0.9, 6.0, 0, d ** -0.5)
    c = ops.init_table(n, d, 5, 0, 0.5, ld=ld)
    c_h = c.cpu().numpy().copy()
    ops.stats_reset(g)
    for part in range(parts):
        x = ops.init_table_rows(stripe_rows(n, part, parts), d, 5, 1, 0.5, part, parts, ld=ld)
        x_h = x.cpu().numpy().copy()
        x0 = x_h.copy()
        ops.block_step(g, tp, plan, pairs, offs, None, None, c, x, 0, part, 5, 0, 0.05)
        O.block_step(og, otp, oplan, words_h, off_h, None, None, c_h, x_h, 0, part, 5, 0, 0.05)
        torch.cuda.synchronize()
        assert np.abs(x.cpu().numpy() - x_h).max() < 2e-5
        assert np.abs(x_h - x0).max() > 1e-3
    st = ops.stats_read(g)
    assert st["resident_launches"] == parts and st["pairs"] == len(words_h)
    assert np.abs(c.cpu().numpy() - c_h).max() < 2e-5


@pytest.mark.parametrize("lpt", ["1", "0"])
def test_cells_taken_heaviest_first_train_the_same_pairs(monkeypatch, lpt):
    """A resident launch of at least two cells per CU hands its cells to the workgroups by
    descending number of pairs (gn2v_block_api.hip lpt_order, GN2V_RESIDENT_LPT=0: index order):
    cells of very different sizes, some empty, collision-free pairs -- every pair trained once,
    tables equal to the sequential oracle either way."""
    monkeypatch.setenv("GN2V_RESIDENT_LPT", lpt)
    parts, slices, record, d = 2, 640, 16, 64
    n = 160_005
    g = _ba(n)
    og = O.OracleGraph(g.row_ptr, g.col_idx)
    plan = ops.block_plan(g, 1, 0, parts, slices, 8, 2, 1, record)
    oplan = O.block_plan(n, 1, 0, parts, slices, 8, 2, 1, record)
    rng = np.random.RandomState(11)
    words_l, offsets, centre = [], [0], 0
    for cell in range(parts * slices):
        part, slc = divmod(cell, slices)
        rows = stripe_rows(stripe_rows(n, part, parts), slc, slices)
        per_cell = 0 if cell % 7 == 3 else int(rng.randint(1, min(rows, 120)))
        ctx = slc + slices * rng.permutation(rows)[:per_cell]
        words_l.append(O.block_pack(np.full(per_cell, cell), centre + np.arange(per_cell), ctx, oplan))
        centre += per_cell
        offsets.append(offsets[-1] + per_cell)
    words_h = np.concatenate(words_l).astype(np.uint64)
    off_h = np.asarray(offsets, dtype=np.uint64)
    pairs, offs = _dev_words(words_h), torch.from_numpy(off_h.astype(np.int64)).cuda()
    tp = ops.train_params(0, d, 0, 2, flags=0, ld=d)
    otp = O.TrainParams(0, d, d, 1, 0, 2, 0.01, 0.9, 6.0, 0, d ** -0.5)
    c = ops.init_table(n, d, 5, 0, 0.5, ld=d)
    c_h = c.cpu().numpy().copy()
    ops.stats_reset(g)
    for part in range(parts):
        x = ops.init_table_rows(stripe_rows(n, part, parts), d, 5, 1, 0.5, part, parts, ld=d)
        x_h = x.cpu().numpy().copy()
        ops.block_step(g, tp, plan, pairs, offs, None, None, c, x, 0, part, 5, 0, 0.05)
        O.block_step(og, otp, oplan, words_h, off_h, None, None, c_h, x_h, 0, part, 5, 0, 0.05)
        torch.cuda.synchronize()
        assert np.abs(x.cpu().numpy() - x_h).max() < 2e-5
    st = ops.stats_read(g)
    assert st["resident_launches"] == parts and st["pairs"] == len(words_h)
    assert np.abs(c.cpu().numpy() - c_h).max() < 2e-5


@pytest.mark.parametrize("d", [16, 128, 256, 384])
def test_parallel_pair_per_group_adds_shared_centres_with_atomics_in_resident_cells(d):
    """Every fourth pair shares its centre with its neighbour (three runs per four pairs: the
    pair-per-group loop) -- both inside one step of four, so both groups read the row before
    either adds its gradient, exactly the oracle's run of two -- unique context rows, k = 0:
    the two gradients must both arrive (f32 atomics through the transposition row)."""
    parts, slices, record = 2, 1024 if d <= 256 else 2048, 32  # (66-row cells at 512 floats)
    n = 160_005
    g = _ba(n)
    og = O.OracleGraph(g.row_ptr, g.col_idx)
    ld = (d + 31) // 32 * 32 if d > 16 else d
    plan = ops.block_plan(g, 1, 0, parts, slices, 8, 2, 1, record)
    oplan = O.block_plan(n, 1, 0, parts, slices, 8, 2, 1, record)
    rng = np.random.RandomState(8)
    per_cell = 64 if d <= 256 else 36
    words_l, offsets, centre = [], [0], 0
    for cell in range(parts * slices):
        part, slc = divmod(cell, slices)
        rows = stripe_rows(stripe_rows(n, part, parts), slc, slices)
        ctx = slc + slices * rng.permutation(rows)[:per_cell]
        steps = np.arange(per_cell) // 4
        centres = centre + steps * 3 + np.maximum(np.arange(per_cell) % 4 - 1, 0)
        centre = int(centres[-1]) + 1
        words_l.append(O.block_pack(np.full(per_cell, cell), centres, ctx, oplan))
        offsets.append(offsets[-1] + per_cell)
    words_h = np.concatenate(words_l).astype(np.uint64)
    off_h = np.asarray(offsets, dtype=np.uint64)
    pairs, offs = _dev_words(words_h), torch.from_numpy(off_h.astype(np.int64)).cuda()
    tp = ops.train_params(0, d, 0, 2, flags=0, ld=ld)
    otp = O.TrainParams(0, d, ld, 1, 0, 2, 0.01, 0.9, 6.0, 0, d ** -0.5)
    c = ops.init_table(n, d, 5, 0, 0.5, ld=ld)
    c_h = c.cpu().numpy().copy()
    ops.stats_reset(g)
    for part in range(parts):
        x = ops.init_table_rows(stripe_rows(n, part, parts), d, 5, 1, 0.5, part, parts, ld=ld)
        x_h = x.cpu().numpy().copy()
        ops.block_step(g, tp, plan, pairs, offs, None, None, c, x, 0, part, 5, 0, 0.1)
        O.block_step(og, otp, oplan, words_h, off_h, None, None, c_h, x_h, 0, part, 5, 0, 0.1)
        torch.cuda.synchronize()
        assert np.abs(x.cpu().numpy() - x_h).max() < 2e-5
    st = ops.stats_read(g)
    assert st["resident_launches"] == parts
    assert st["pairs"] == len(words_h) == st["centres"]  # one hand-over per pair: that loop
    assert np.abs(c.cpu().numpy() - c_h).max() < 2e-5


def test_a_group_of_parts_in_one_deterministic_launch_equals_part_by_part():
    """gn2v_block_round hands a whole extraction group to ONE launch of the resident kernel; its
    deterministic form walks the parts itself.  Through the round driver (BlockPartitionedTrainer
    on one GPU) with the deterministic flag: equal to the oracle-backed trainer."""
    from embiggen_amd.distributed import BlockPartitionedTrainer, LoopbackComm
    from sharded_helpers import OracleBlockBackend

    n, d, k, w, L = 1999, 32, 4, 3, 14
    g = _ba(n)
    og = O.OracleGraph(g.row_ptr, g.col_idx)
    wk_h = O.walks(og, O.WalkParams(L, 2, 0.5, 2.0, 100, 0), 7, 0, 0, 500)
    out = []
    for use_oracle in (False, True):
        if use_oracle:
            tp = O.TrainParams(0, d, d, 1, k, w, 0.05, 0.9, 6.0, 1, d ** -0.5)
            tr = BlockPartitionedTrainer(g, tp, d, d, 7, d ** -0.5, LoopbackComm(), "cpu",
                                         walk_length=L, window=w, parts=4, slices=24, record=16,
                                         backend=OracleBlockBackend(g), group_parts=2)
            wk = torch.from_numpy(wk_h.view(np.int32))
        else:
            tp = ops.train_params(0, d, k, w, flags=1 | DET, ld=d)
            tr = BlockPartitionedTrainer(g, tp, d, d, 7, d ** -0.5, LoopbackComm(), "cuda:0",
                                         walk_length=L, window=w, parts=4, slices=24, record=16,
                                         group_parts=2)
            wk = torch.from_numpy(wk_h.view(np.int32)).cuda()
            ops.stats_reset(g)
        tr.train_round(wk, 7, 0, 0.05, 0)
        if not use_oracle:
            torch.cuda.synchronize()
            st = ops.stats_read(g)
            assert st["resident_launches"] == 2, st  # two groups of two parts, one launch each
        out.append(tuple(t.cpu().numpy() for t in tr.gather_full()))
    (c, x), (c_h, x_h) = out
    assert np.abs(c - c_h).max() < 1e-5 and np.abs(x - x_h).max() < 1e-5


# ---------------------------------------------------------------- what the parallel form loses
def _in_cell_displacement(top_share, n_pairs=200_000, d=128, rows=200, flags=0):
    """ONE resident cell of ``rows`` rows, n_pairs pairs with unique centres, k = 0, a learning
    rate so small that the order of the updates does not matter: row 0 is the context of
    ``top_share`` of the pairs, the others share the rest evenly.  Returns how far row 0 and the
    median other row moved, relative to the sequential oracle."""
    parts, slices = 1, 1024
    n = rows * slices
    assert n > n_pairs
    g = E.barabasi_albert(n, 3, 9)
    og = O.OracleGraph(g.row_ptr, g.col_idx)
    plan = ops.block_plan(g, 1, 0, parts, slices, 8, 2, 1, 32)
    oplan = O.block_plan(n, 1, 0, parts, slices, 8, 2, 1, 32)
    cell = 5
    cell_rows = stripe_rows(stripe_rows(n, 0, parts), cell, slices)
    assert cell_rows >= rows
    rng = np.random.RandomState(1)
    local = np.where(rng.rand(n_pairs) < top_share, 0, 1 + rng.randint(0, rows - 1, n_pairs))
    ctx = cell + slices * local
    words_h = O.block_pack(np.full(n_pairs, cell), np.arange(n_pairs), ctx, oplan).astype(np.uint64)
    off_h = np.zeros(parts * slices + 1, dtype=np.uint64)
    off_h[cell + 1:] = n_pairs
    pairs, offs = _dev_words(words_h), torch.from_numpy(off_h.astype(np.int64)).cuda()
    lr = 1e-8
    tp = ops.train_params(0, d, 0, 2, flags=flags, ld=d)
    otp = O.TrainParams(0, d, d, 1, 0, 2, 0.01, 0.9, 6.0, 0, d ** -0.5)
    c = ops.init_table(n, d, 5, 0, 0.5).abs_()  # all positive: the gradients add up
    x = torch.zeros(n, d, device="cuda")
    c_h, x_h = c.cpu().numpy().copy(), x.cpu().numpy().copy()
    ops.stats_reset(g)
    ops.block_step(g, tp, plan, pairs, offs, None, None, c, x, 0, 0, 5, 0, lr)
    O.block_step(og, otp, oplan, words_h, off_h, None, None, c_h, x_h, 0, 0, 5, 0, lr)
    torch.cuda.synchronize()
    assert _resident_launches(g) == 1
    got, want = x.cpu().numpy(), x_h
    used = np.unique(ctx)
    ratio = np.array([np.median(got[r] / want[r]) for r in used])
    top = float(ratio[0]) if used[0] == cell else float("nan")
    rest = ratio[1:] if len(ratio) > 1 else np.array([np.nan])
    return top, float(np.median(rest)), float(rest.min())


def test_in_cell_losses_are_counted():
    """How much of a row's movement survives the workgroup's own races (VERDICT r4
    weak 4) -- printed, and bounded from below so that a regression shows: the extreme (every pair
    of the cell on one row), the realistic hub (the top row of a bench-graph cell receives ~4 %
    of its samples) and an even spread.  Measured, round 5 (profiles/r05_logs/): round 4's kernel
    0.907 / 0.988 (hub row / ordinary rows); two thirds of the hub row's loss were its own wave's
    four groups naming it in the same instruction, which the staged phases now serialise: 0.919 /
    0.989 at a kernel 15 % faster.  What is left is another WAVE's update landing between a
    row's second read and its stores.  Exact alternatives, measured: ds_add_f32 on every element
    costs 24 x a plain read-modify-write (scripts/lds_atomic_probe.hip: 6.1e6 against 1.5e8 row
    updates per second and CU), a per-row lock serialises the hub rows (round 4)."""
    report = {}
    for name, share in (("every pair on one row", 1.0), ("top row 4 %", 0.04),
                        ("top row 1 %", 0.01), ("even", 0.005)):
        report[name] = _in_cell_displacement(share)
    print("resident cell, displacement / sequential (top row, median other row, worst other row):")
    for name, r in report.items():
        print(f"  {name:24s} {r[0]:.4f} {r[1]:.4f} {r[2]:.4f}")
    assert report["top row 4 %"][0] >= 0.89, report
    assert report["top row 1 %"][0] >= 0.96, report
    assert report["top row 4 %"][1] >= 0.98 and report["even"][1] >= 0.98, report
    assert report["even"][2] >= 0.96, report


# ------------------------------------------------------------ placement of a round (round 5)
@pytest.mark.parametrize("n,classes", [(34, 1), (1999, 1), (1999, 6), (100_003, 1), (100_003, 16)])
def test_placement_is_bit_equal_to_the_oracles(n, classes):
    """gn2v_block_placement (splitmix64 keys, rocPRIM radix sort with ties in node order, one
    scatter) against o_block_placement (qsort on (key, node)): the same permutation, bit for bit,
    and gn2v_block_place_walks maps walks through it."""
    g = E.karate_club() if n == 34 else E.barabasi_albert(n, 3, 9)
    for seed, rnd in ((42, 0), (42, 7), (5, 123456789)):
        place, inv = ops.block_placement(g, classes, seed, rnd)
        rp, ri = O.block_placement(n, classes, seed, rnd)
        assert np.array_equal(place.cpu().numpy().view(np.uint32), rp)
        assert np.array_equal(inv.cpu().numpy().view(np.uint32), ri)
    wk = ops.walks(g, ops.walk_params(12, 1, 1.0, 1.0), 3, 0, 0, min(n, 500))
    wk[::7, 5:] = -1
    placed = ops.block_place_walks(place, wk).cpu().numpy().view(np.uint32)
    w = wk.cpu().numpy().view(np.uint32)
    assert np.array_equal(placed[w != O.SENTINEL], rp[w[w != O.SENTINEL]])
    assert (placed[w == O.SENTINEL] == O.SENTINEL).all()


@pytest.mark.parametrize("classes,parts,slices,natural", [(1, 2, 32, True), (1, 3, 17, True),
                                                          (4, 4, 24, False)])
@pytest.mark.parametrize("d,k", [(16, 4), (128, 10)])
def test_deterministic_resident_step_under_a_placement_matches_oracle(classes, parts, slices,
                                                                      natural, d, k):
    """A round under a placement, piece by piece against the oracle: alias tables through the
    placement's inverse and pair words from the placed walks bit-exact, then the deterministic
    resident step reaching the rows through the inverse -- in the whole table in node order
    (classes = 1: one GPU) or in the parts' own buffers (classes = parts: several ranks)."""
    n, wl, window, record = 1999, 20, 4, 16
    g = _ba(n)
    og = O.OracleGraph(g.row_ptr, g.col_idx)
    ld = (d + 31) // 32 * 32 if d > 16 else d
    place, inv = ops.block_placement(g, classes, 11, 5)
    rp, ri = O.block_placement(n, classes, 11, 5)
    plan = ops.block_plan(g, 1, 0, parts, slices, wl, window, 1, record)
    oplan = O.block_plan(n, 1, 0, parts, slices, wl, window, 1, record)
    alias, cell_rows = ops.block_alias(g, plan, inv=inv)[:2]
    ra, rc_rows = O.block_alias(og, parts, slices, 0, inv=ri)[:2]
    assert np.array_equal(alias.cpu().numpy().view(np.uint64), ra)
    assert not np.array_equal(ra, O.block_alias(og, parts, slices, 0)[0])  # another grouping
    wk = ops.walks(g, ops.walk_params(wl, 2, 0.5, 2.0), 11, 0, 0, 400)
    wk[::9, 11:] = -1
    placed = ops.block_place_walks(place, wk)
    work, offsets = ops.block_count(g, plan, wk, 11, 0, 0, placed=placed)
    pairs = ops.block_extract(g, plan, wk, 11, 0, 0, work, int(offsets[-1]), placed=placed)
    rw, ro = O.block_extract(og, oplan, wk.cpu().numpy().view(np.uint32), 11, 0, 0, place=rp)
    assert np.array_equal(_words(pairs), rw) and len(rw) > 10_000
    assert np.array_equal(offsets.cpu().numpy().astype(np.uint64), ro)
    tp = ops.train_params(0, d, k, window, flags=1 | DET, ld=ld)
    otp = O.TrainParams(0, d, ld, 1, k, window, 0.01, 0.9, 6.0, 1, d ** -0.5)
    c = ops.init_table(n, d, 11, 0, d ** -0.5, ld=ld)
    x = ops.init_table(n, d, 11, 1, d ** -0.5, ld=ld)
    c_h, x_h = c.cpu().numpy().copy(), x.cpu().numpy().copy()
    x0 = x_h.copy()
    ops.stats_reset(g)
    for part in range(parts):
        if natural:
            ops.block_step(g, tp, plan, pairs, offsets, alias, cell_rows, c, None, 3, part, 11, 0,
                           0.05, inv=inv, context_table=x)
        else:  # the part's own buffer, as it travels between ranks
            rows = x[part::parts].contiguous()
            ops.block_step(g, tp, plan, pairs, offsets, alias, cell_rows, c, rows, 3, part, 11, 0,
                           0.05, inv=inv)
            x[part::parts] = rows
        O.block_step(og, otp, oplan, rw, ro, ra, rc_rows, c_h, x_h, 3, part, 11, 0, 0.05, inv=ri,
                     natural=True)
    torch.cuda.synchronize()
    st = ops.stats_read(g)
    assert st["resident_launches"] == parts and st["pairs"] == len(rw)
    assert np.abs(c.cpu().numpy() - c_h).max() < 1e-5
    assert np.abs(x.cpu().numpy() - x_h).max() < 1e-5
    assert np.abs(x_h - x0).max() > 1e-3


def test_parallel_resident_step_under_a_placement_on_collision_free_pairs():
    """The shipping (parallel) form reaching its rows through a placement's inverse, every centre
    and every context row used once, k = 0: equal to the sequential oracle."""
    parts, slices, record, d, per_cell = 2, 512, 16, 128, 150
    n = 160_005
    g = E.barabasi_albert(n, 3, 9)
    og = O.OracleGraph(g.row_ptr, g.col_idx)
    place, inv = ops.block_placement(g, 1, 3, 9)
    ri = inv.cpu().numpy().view(np.uint32)
    plan = ops.block_plan(g, 1, 0, parts, slices, 8, 2, 1, record)
    oplan = O.block_plan(n, 1, 0, parts, slices, 8, 2, 1, record)
    rng = np.random.RandomState(3)
    words_l, offsets, centre = [], [0], 0
    for cell in range(parts * slices):
        part, slc = divmod(cell, slices)
        rows = stripe_rows(stripe_rows(n, part, parts), slc, slices)
        ctx = slc + slices * rng.permutation(rows)[:per_cell]
        words_l.append(O.block_pack(np.full(per_cell, cell), centre + np.arange(per_cell), ctx, oplan))
        centre += per_cell
        offsets.append(offsets[-1] + per_cell)
    words_h = np.concatenate(words_l).astype(np.uint64)
    off_h = np.asarray(offsets, dtype=np.uint64)
    pairs, offs = _dev_words(words_h), torch.from_numpy(off_h.astype(np.int64)).cuda()
    tp = ops.train_params(0, d, 0, 2, flags=0, ld=d)
    otp = O.TrainParams(0, d, d, 1, 0, 2, 0.01, 0.9, 6.0, 0, d ** -0.5)
    c = ops.init_table(n, d, 5, 0, 0.5)
    x = ops.init_table(n, d, 5, 1, 0.5)
    c_h, x_h = c.cpu().numpy().copy(), x.cpu().numpy().copy()
    for part in range(parts):
        ops.block_step(g, tp, plan, pairs, offs, None, None, c, None, 0, part, 5, 0, 0.05,
                       inv=inv, context_table=x)
        O.block_step(og, otp, oplan, words_h, off_h, None, None, c_h, x_h, 0, part, 5, 0, 0.05,
                     inv=ri, natural=True)
    torch.cuda.synchronize()
    assert np.abs(x.cpu().numpy() - x_h).max() < 2e-5
    assert np.abs(c.cpu().numpy() - c_h).max() < 2e-5


@pytest.mark.parametrize("world,parts,slices", [(1, 3, 24), (2, 4, 24)])
def test_trainer_under_a_placement_equals_the_oracle_backed_trainer(world, parts, slices):
    """BlockPartitionedTrainer on a plan of resident cells (a placement per round: over the whole
    graph on one rank -- through the C round driver --, inside the classes modulo `parts` with
    two simulated ranks -- the per-group loop, parts travelling) in the deterministic form,
    against the same trainer computing with the oracle."""
    from embiggen_amd.distributed import BlockPartitionedTrainer
    from sharded_helpers import OracleBlockBackend, run_ranks

    n, d, k, w, L, wpr = 1999, 16, 4, 3, 14, 150
    g = _ba(n)
    og = O.OracleGraph(g.row_ptr, g.col_idx)

    def run(comm, use_oracle):
        if use_oracle:
            tp = O.TrainParams(0, d, d, 1, k, w, 0.05, 0.9, 6.0, 1, d ** -0.5)
            tr = BlockPartitionedTrainer(g, tp, d, d, 7, d ** -0.5, comm, "cpu", walk_length=L,
                                         window=w, parts=parts, slices=slices, record=16,
                                         backend=OracleBlockBackend(g), group_parts=2)
        else:
            tp = ops.train_params(0, d, k, w, flags=1 | DET, ld=d)
            tr = BlockPartitionedTrainer(g, tp, d, d, 7, d ** -0.5, comm, "cuda:0",
                                         walk_length=L, window=w, parts=parts, slices=slices,
                                         record=16, group_parts=2)
        assert tr.permute and tr.natural == (comm.world == 1)
        for r in range(2):
            first = r * comm.world * wpr
            mine = O.walks(og, O.WalkParams(L, 2, 0.5, 2.0, 100, 0), 7, 0, first + comm.rank * wpr,
                           wpr)
            wk = torch.from_numpy(mine.view(np.int32))
            tr.train_round(wk if use_oracle else wk.cuda(), 7, 0, 0.05, first)
        return [t.cpu().numpy() for t in tr.gather_full()]

    gpu = run_ranks(world, lambda comm: run(comm, False))
    ref = run_ranks(world, lambda comm: run(comm, True))
    for r in range(world):
        assert np.abs(gpu[r][0] - ref[r][0]).max() < 1e-5
        assert np.abs(gpu[r][1] - ref[r][1]).max() < 1e-5
    assert np.abs(ref[0][1] - O.init_table(n, d, d, 7, 1, d ** -0.5)).max() > 1e-3


def test_gn2v_train_takes_resident_cells_under_a_placement_and_equals_the_python_trainer(monkeypatch):
    """The C++ fit (gn2v_train -> gn2v_train_blocks: automatic plan, a placement + alias tables +
    placed walks per round, the contextual table in node order) against the Python trainer making
    the same calls (which the test above ties to the oracle), deterministic form, two epochs of
    two rounds, on a graph small enough for one workgroup to walk it (GN2V_RESIDENT_MIN_NODES
    lowered for the occasion)."""
    from embiggen_amd.distributed import LoopbackComm

    monkeypatch.setenv("GN2V_RESIDENT_MIN_NODES", "3000")
    monkeypatch.setenv("GN2V_RESIDENT_MAX_SKEW_PCT", "100000")  # a graph this small is all hub
    g = E.barabasi_albert(6000, 3, 9)
    kw = dict(embedding_size=100, epochs=2, walk_length=12, iterations=1, window_size=3,
              number_of_negative_samples=4, learning_rate=0.05, learning_rate_decay=0.8,
              return_weight=2.0, explore_weight=0.5, deterministic=True, verbose=False)
    m_c = E.models.SkipGram(block_path=True, **kw)
    c1, x1, st = m_c.fit_transform_device(g, max_walks_per_epoch=700)
    assert st["block_slices"] > 16 and st["resident_launches"] > 0, st
    m_py = E.models.SkipGram(**kw)
    c2, x2 = m_py.fit_transform_blocks(g, LoopbackComm(), max_walks_per_epoch=700, round_walks=400)
    assert m_py.last_plan["slices"] == st["block_slices"] and m_py.last_plan["parts"] == st["block_parts"]
    assert st["pairs"] == m_py.last_stats["pairs"] == 2 * 700 * (2 * 3 * 12 - 3 * 4)
    # (gn2v_train sized its rounds itself: the same rounds only when it took 400 walks too)
    if st["block_round_walks"] == 400:
        assert float((c1 - c2).abs().max()) < 1e-5 and float((x1 - x2).abs().max()) < 1e-5
    c3, x3 = E.models.SkipGram(**kw).fit_transform_blocks(g, LoopbackComm(), max_walks_per_epoch=700,
                                                          round_walks=st["block_round_walks"])
    assert float((c1 - c3).abs().max()) < 1e-5 and float((x1 - x3).abs().max()) < 1e-5
    init = ops.init_table(6000, 100, 42, 1, 100 ** -0.5, ld=128)
    assert float((x1[:, :100] - init[:, :100]).abs().max()) > 1e-3


def test_negatives_met_by_a_context_over_the_rounds():
    """What the placement is for, measured on the device.  One fixed context
    x0; every round it is placed in another cell of 200 rows, and 300 pairs (c_i, x0) draw
    k = 5 negatives each there -- deterministic form, central rows = one constant vector,
    contextual rows = 0, a tiny learning rate: row y ends at -(times y was drawn) lr / 2 u.
    (a) GIVEN the cell-mates of every round the counts follow in-degree / cell total (chi-square,
        the skip rule for negative == context included);
    (b) over the rounds x0 meets negatives from the whole graph: after 96 rounds ~60 % of all
        nodes have been cell-mates, every one about equally often, where fixed cells would have
        offered the same 199 nodes 96 times;
    (c) per degree class the counts stay within 6 % of in-degree / total for the classes that
        hold 90 % of the in-degree (measured 0.98-1.02); the class of the hubs gets 0.97 here
        -- a cell normalises by its OWN total, so a hub that is a mate crowds the very draw it
        is in (a node whose in-degree is a tenth of a cell's total is drawn 9 % less often than
        the law over the graph says): printed, and bounded."""
    from scipy import stats

    n, d, k, slices, rounds, n_pairs = 20_000, 8, 5, 100, 96, 300
    g = E.barabasi_albert(n, 5, 42)
    og = O.OracleGraph(g.row_ptr, g.col_idx)
    indeg = np.bincount(g.col_idx, minlength=n).astype(np.float64)
    plan = ops.block_plan(g, 1, 0, 1, slices, 8, 2, 1, 16)
    oplan = O.block_plan(n, 1, 0, 1, slices, 8, 2, 1, 16)
    x0 = 12_345
    centres = np.arange(1, n_pairs + 1) * 17 % n
    centres = centres[centres != x0]
    lr, u = 1e-6, d ** -0.5
    tp = ops.train_params(0, d, k, 2, flags=1 | DET, ld=d)
    c = torch.full((n, d), u, device="cuda")
    x = torch.zeros((n, d), device="cuda")
    expected = np.zeros(n)
    met = np.zeros(n, dtype=np.int64)
    for r in range(rounds):
        place, inv = ops.block_placement(g, 1, 42, r)
        alias, cell_rows = ops.block_alias(g, plan, inv=inv)[:2]
        xp = int(place[x0])  # parts = 1: row = x', slice = row % slices, local = row // slices
        cell, local = xp % slices, xp // slices
        words = O.block_pack(np.full(len(centres), cell), centres, np.full(len(centres), xp), oplan)
        off = np.zeros(slices + 1, dtype=np.uint64)
        off[cell + 1:] = len(centres)
        ops.block_step(g, tp, plan, _dev_words(np.sort(words)), torch.from_numpy(off.astype(np.int64)).cuda(),
                       alias, cell_rows, c, None, r, 0, 42, 0, lr, inv=inv, context_table=x)
        inv_h = inv.cpu().numpy().view(np.uint32)
        rows_in_cell = -(-(n - cell) // slices)
        mates = inv_h[cell + slices * np.arange(rows_in_cell)].astype(np.int64)
        assert x0 in mates and mates[local] == x0
        w = indeg[mates]
        # a draw of x0 itself is skipped; a draw of the pair's own centre too (one pair in 299)
        share = w / w.sum()
        share[local] = 0.0
        expected[mates] += len(centres) * k * share
        met[mates] += 1
    torch.cuda.synchronize()
    got = (-x.double().mean(1) / (0.5 * lr * u)).cpu().numpy()
    got[x0] = 0.0
    counts = np.rint(got)
    # (a row that was drawn 500 times has moved far enough for its sigmoid to leave 1 / 2 a little)
    assert (np.abs(got - counts) <= 0.05 + 2e-3 * counts).all() and counts.min() >= 0
    assert abs(counts.sum() / expected.sum() - 1) < 0.01
    # (a) given the mates: chi-square over the nodes ever met, pooled to >= 40 expected draws
    order = np.argsort(-expected)
    order = order[expected[order] > 0]
    bins_o, bins_e, acc_o, acc_e = [], [], 0.0, 0.0
    for y in order:
        acc_o += counts[y]
        acc_e += expected[y]
        if acc_e >= 40:
            bins_o.append(acc_o)
            bins_e.append(acc_e)
            acc_o = acc_e = 0.0
    o, e = np.array(bins_o), np.array(bins_e)
    p = stats.chisquare(o, e * o.sum() / e.sum()).pvalue
    assert len(o) > 100 and p > 1e-4, p
    # (b) the mates over the rounds
    others = np.delete(np.arange(n), x0)
    assert (met[others] > 0).mean() > 0.55 and met[others].max() <= 8
    assert abs(met[others].mean() - rounds * 199 / (n - 1)) < 0.05
    # (c) against the law over the whole graph, by degree class (equal shares of the in-degree)
    by_deg = np.argsort(indeg)
    cum = np.cumsum(indeg[by_deg]) / indeg.sum()
    report = []
    for lo, hi in ((0.0, 0.3), (0.3, 0.6), (0.6, 0.9), (0.9, 1.0)):
        cls = by_deg[(cum > lo) & (cum <= hi)]
        cls = cls[cls != x0]
        report.append((float(indeg[cls].min()), float(indeg[cls].max()),
                       float(counts[cls].sum() / counts.sum()), float(indeg[cls].sum() / indeg[others].sum())))
    print("degree class (min, max in-degree): share of the negatives met / share of the in-degree")
    for lo_d, hi_d, got_s, want_s in report:
        print(f"  {lo_d:6.0f} .. {hi_d:6.0f}: {got_s:.4f} / {want_s:.4f} = {got_s / want_s:.3f}")
    # measured: 0.983, 1.009, 1.018, 0.969
    for lo_d, hi_d, got_s, want_s in report[:3]:
        assert abs(got_s / want_s - 1) < 0.06, report
    assert 0.88 < report[3][2] / report[3][3] < 1.08, report

"""GPU: pair extraction, the pair-mode kernel and the block-partitioned multi-GPU trainer with
ranks simulated on one GPU (exact: ranks never share a row, so a sequential simulation computes
precisely what W GPUs compute; only the transport differs)."""
import numpy as np
import pytest
import torch

import embiggen_amd as E
from embiggen_amd import _lib, ops
from embiggen_amd.distributed import BlockPartitionedTrainer
from oracle import oracle as O
from sharded_helpers import (host_init_fn, host_walk_pair_blocks, link_auc_device as _auc,
                             oracle_block_compute, run_ranks)

pytestmark = pytest.mark.gpu
D, K, W, L = 16, 4, 3, 14


def test_walk_pairs_match_oracle(karate, karate_oracle):
    wk = ops.walks(karate, ops.walk_params(20, 2, 0.5, 2.0), 3, 0, 0, 68)
    wk_h = wk.cpu().numpy().view(np.uint32)
    for window, md in ((3, 1), (5, 1), (4, 4), (4, 2)):
        got = ops.walk_pairs(wk, window, md).cpu().numpy().view(np.uint32)
        assert np.array_equal(got, O.walk_pairs(wk_h, window, md))
    # trap nodes: sentinel suffixes produce no pairs
    cut = wk.clone()
    cut[:, 7:] = -1
    got = ops.walk_pairs(cut, 3).cpu().numpy().view(np.uint32)
    assert np.array_equal(got, O.walk_pairs(cut.cpu().numpy().view(np.uint32), 3))


@pytest.mark.parametrize("flags", [_lib.TRAIN_DETERMINISTIC, _lib.TRAIN_ATOMIC,
                                   _lib.TRAIN_WRITE_THROUGH])
def test_pair_mode_step_matches_oracle(karate, karate_oracle, flags):
    wk = ops.walks(karate, ops.walk_params(10, 1, 1.0, 1.0), 4, 0, 0, 34)
    pairs = ops.walk_pairs(wk, 2)
    pairs_h = pairs.cpu().numpy().view(np.uint32)
    c, x = ops.init_table(34, D, 4, 0, D ** -0.5), ops.init_table(34, D, 4, 1, D ** -0.5)
    c_h, x_h = c.cpu().numpy().copy(), x.cpu().numpy().copy()
    tp = ops.train_params(0, D, K, 1, flags=1 | flags)
    otp = O.TrainParams(0, D, D, 1, K, 1, 0.01, 0.9, 6.0, 1, D ** -0.5)
    if flags == _lib.TRAIN_DETERMINISTIC:
        ops.step(karate, tp, pairs, 4, 0, 100, 0.05, c, x, pair_mode=True)
    else:  # production flavours one pair per launch (pairs of a batch share rows)
        for b in range(0, 200):
            ops.step(karate, tp, pairs[b:b + 1].contiguous(), 4, 0, 100 + b, 0.05, c, x,
                     pair_mode=True)
        pairs_h = pairs_h[:200]
    torch.cuda.synchronize()
    O.train_walks_ex(karate_oracle, otp, pairs_h, 4, 0, 100, 0.05, c_h, x_h, pair_mode=True)
    assert np.abs(c.cpu().numpy() - c_h).max() < 1e-5 and np.abs(x.cpu().numpy() - x_h).max() < 1e-5


def _run(comm, device, use_oracle, graph=None):
    g = E.karate_club() if graph is None else graph
    og = O.OracleGraph(g.row_ptr, g.col_idx)
    n_nodes = g.get_number_of_nodes()
    otp = O.TrainParams(0, D, D, 1, K, 1, 0.02, 0.9, 6.0, 1, D ** -0.5)
    tp = ops.train_params(0, D, K, 1, flags=1 | _lib.TRAIN_DETERMINISTIC)
    if use_oracle:
        tr = BlockPartitionedTrainer(g, otp, D, D, 42, D ** -0.5, comm, "cpu",
                                     init_fn=host_init_fn(n_nodes, D, D, 42, D ** -0.5))
        tr.compute = oracle_block_compute(og, otp, tr)
    else:
        tr = BlockPartitionedTrainer(g, tp, D, D, 42, D ** -0.5, comm, device)
    wp = O.WalkParams(L, 1, 0.25, 4.0, 100, 0)
    for r in range(2):
        first = (r * comm.world + comm.rank) * 9
        walks = torch.from_numpy(O.walks(og, wp, 42, 0, first, 9).view(np.int32)).to(device)
        # explicit pairs on both sides: the same (torch-built) shuffle keys, hence the same order
        pairs = torch.from_numpy(
            O.walk_pairs(walks.cpu().numpy().view(np.uint32), W).view(np.int32)).to(device)
        tr.train_round(None, W, 1, 42, 0, 0.02, pairs=pairs)
    return [t.cpu().numpy() for t in tr.gather_full()]


@pytest.mark.parametrize("world", [1, 2, 3])
def test_block_trainer_kernel_equals_oracle(world):
    gpu = run_ranks(world, lambda comm: _run(comm, "cuda:0", use_oracle=False))
    ref = run_ranks(world, lambda comm: _run(comm, "cpu", use_oracle=True))
    for r in range(world):
        assert np.abs(gpu[r][0] - ref[r][0]).max() < 1e-5
        assert np.abs(gpu[r][1] - ref[r][1]).max() < 1e-5


@pytest.mark.parametrize("world,nodes", [(4, 203), (5, 97), (8, 64)])
def test_block_trainer_on_scale_free_graphs_with_ragged_partitions(world, nodes):
    """Partitions of unequal size (nodes % world != 0), hubs in one partition, more ranks."""
    s, d = O.ba_edges(nodes, 3, 9)
    g = E.CSRGraph.from_edge_list(s, d, number_of_nodes=nodes)
    gpu = run_ranks(world, lambda comm: _run(comm, "cuda:0", use_oracle=False, graph=g))
    ref = run_ranks(world, lambda comm: _run(comm, "cpu", use_oracle=True, graph=g))
    for r in range(world):
        assert gpu[r][0].shape == (nodes, D)
        assert np.abs(gpu[r][0] - ref[r][0]).max() < 1e-5
        assert np.abs(gpu[r][1] - ref[r][1]).max() < 1e-5
    assert all(np.array_equal(gpu[0][0], gpu[r][0]) for r in range(world))  # gather_full agrees


def _run_fused(comm, device, use_oracle, graph, contexts):
    """The production route: walks in, pairs grouped by (block, centre) and packed into centre
    records, records routed and trained."""
    og = O.OracleGraph(graph.row_ptr, graph.col_idx)
    n_nodes = graph.get_number_of_nodes()
    otp = O.TrainParams(0, D, D, 1, K, 1, 0.02, 0.9, 6.0, 1, D ** -0.5)
    tp = ops.train_params(0, D, K, 1, flags=1 | _lib.TRAIN_DETERMINISTIC)
    if use_oracle:
        tr = BlockPartitionedTrainer(graph, otp, D, D, 42, D ** -0.5, comm, "cpu",
                                     init_fn=host_init_fn(n_nodes, D, D, 42, D ** -0.5),
                                     record_contexts=contexts)
        tr.compute = oracle_block_compute(og, otp, tr)
    else:
        tr = BlockPartitionedTrainer(graph, tp, D, D, 42, D ** -0.5, comm, device,
                                     record_contexts=contexts)
    wp = O.WalkParams(L, 1, 0.25, 4.0, 100, 0)
    infos = []
    for r in range(2):
        first = (r * comm.world + comm.rank) * 11
        walks = torch.from_numpy(O.walks(og, wp, 42, 0, first, 11).view(np.int32)).to(device)
        tr.train_round(walks, W, 1, 42, 0, 0.02)
        infos.append(tr.last_round)
    return [t.cpu().numpy() for t in tr.gather_full()], infos


@pytest.mark.parametrize("world,contexts", [(1, 10), (2, 10), (3, 4), (4, 1)])
def test_fused_centre_record_route_equals_oracle(world, contexts, monkeypatch):
    s, d = O.ba_edges(150, 3, 4)
    g = E.CSRGraph.from_edge_list(s, d, number_of_nodes=150)
    gpu = run_ranks(world, lambda comm: _run_fused(comm, "cuda:0", False, g, contexts))
    monkeypatch.setattr(ops, "walk_pair_blocks", host_walk_pair_blocks)
    ref = run_ranks(world, lambda comm: _run_fused(comm, "cpu", True, g, contexts))
    per_walk = 2 * W * L - W * (W + 1)
    for r in range(world):
        assert np.abs(gpu[r][0][0] - ref[r][0][0]).max() < 1e-5
        assert np.abs(gpu[r][0][1] - ref[r][0][1]).max() < 1e-5
        assert gpu[r][1] == ref[r][1]
    for rnd in range(2):  # every pair of the round is trained exactly once, somewhere
        assert sum(gpu[r][1][rnd]["pairs_trained"] for r in range(world)) == world * 11 * per_walk
        assert all(gpu[r][1][rnd]["pairs_generated"] == 11 * per_walk for r in range(world))


def test_eight_simulated_gpus_reach_single_gpu_quality():
    """BA 200 k nodes: 8 block-partitioned ranks vs one walk-mode trainer on the same walks
    (10 per node).  Merging replica deltas collapses here (AUROC 0.02-0.27);
    orthogonal blocks must stay at the single-GPU quality."""
    g = E.barabasi_albert(200_000, 8, 42)
    n, d, w = g.get_number_of_nodes(), 64, 4
    wp = ops.walk_params(64, 1, 1.0, 1.0)
    total, per_round = 1 << 21, 1 << 15
    c = ops.init_table(n, d, 42, 0, d ** -0.5)
    x = ops.init_table(n, d, 42, 1, d ** -0.5)
    tp_walk = ops.train_params(0, d, 5, w, flags=1)
    for first in range(0, total, per_round):
        ops.sgns_step(g, tp_walk, ops.walks(g, wp, 42, 0, first, per_round), 42, 0, first, 0.025,
                      c, x)
    gen = torch.Generator(device="cuda")
    gen.manual_seed(1)
    auc_single = _auc(g, c, x, gen)

    world = 8
    tp_pair = ops.train_params(0, d, 5, 1, flags=1)

    def rank_fn(comm):
        tr = BlockPartitionedTrainer(g, tp_pair, d, d, 42, d ** -0.5, comm, "cuda:0")
        for r in range(total // per_round // world):
            first = (r * world + comm.rank) * per_round
            tr.train_round(ops.walks(g, wp, 42, 0, first, per_round), w, 1, 42, 0, 0.025)
        return tr.gather_full(), tr.last_round

    (bc, bx), info = run_ranks(world, rank_fn)[0]
    gen.manual_seed(1)
    auc_blocks = _auc(g, bc, bx, gen)
    assert bool(torch.isfinite(bc).all()) and min(info["block_sizes"]) > 0
    assert auc_single > 0.9 and auc_blocks > auc_single - 0.03, (auc_blocks, auc_single)


@pytest.mark.parametrize("flags", [_lib.TRAIN_ATOMIC, _lib.TRAIN_WRITE_THROUGH,
                                   _lib.TRAIN_WRITE_BACK])
@pytest.mark.parametrize("d", [8, 128])
def test_pair_mode_on_a_collision_free_batch(karate, karate_oracle, flags, d):
    """Thousands of (centre, context) records in one launch: with every record on its own rows
    the parallel schedule must equal the sequential oracle."""
    rng = np.random.RandomState(3)
    n_rec, k, blk = 3000, 6, 16
    pairs = np.zeros((n_rec, 2), dtype=np.uint32)
    neg = np.zeros((n_rec, 2, 2, k), dtype=np.uint32)
    for b in range(n_rec):
        nodes = b * blk + rng.permutation(blk)
        pairs[b] = nodes[:2]
        neg[b, 0, 1] = nodes[2:2 + k]
    n_rows = n_rec * blk
    c, x = ops.init_table(n_rows, d, 5, 0, d ** -0.5), ops.init_table(n_rows, d, 5, 1, d ** -0.5)
    c_h, x_h = c.cpu().numpy().copy(), x.cpu().numpy().copy()
    tp = ops.train_params(0, d, k, 1, flags=1 | flags)
    otp = O.TrainParams(0, d, (d + 3) // 4 * 4, 1, k, 1, 0.01, 0.9, 6.0, 1, d ** -0.5)
    ops.step(karate, tp, torch.from_numpy(pairs.view(np.int32)).cuda(), 5, 0, 77, 0.05, c, x,
             neg_override=torch.from_numpy(neg.view(np.int32)).cuda(), pair_mode=True)
    torch.cuda.synchronize()
    O.train_walks_ex(karate_oracle, otp, pairs, 5, 0, 77, 0.05, c_h, x_h, neg_override=neg,
                     pair_mode=True)
    assert np.abs(c.cpu().numpy() - c_h).max() < 1e-5 and np.abs(x.cpu().numpy() - x_h).max() < 1e-5
    assert np.abs(x_h - ops.init_table(n_rows, d, 5, 1, d ** -0.5).cpu().numpy()).max() > 1e-3


def test_model_level_multi_gpu_fit_with_simulated_ranks():
    """``models.SkipGram.fit_transform_blocks``: whole fits (epochs, lr decay, rounds where some
    rank has no walks left) on 1, 2 and 3 simulated ranks give every rank the same full tables and
    the single-GPU quality on a graph with communities."""
    from helpers import link_auc, ring_of_cliques

    src, dst, n = ring_of_cliques(32, 8)
    g = E.CSRGraph.from_edge_list(src, dst, number_of_nodes=n)
    kw = dict(embedding_size=16, epochs=4, walk_length=32, iterations=4, window_size=4,
              number_of_negative_samples=5, learning_rate=0.025, return_weight=1.0,
              explore_weight=1.0, verbose=False)
    single = E.models.SkipGram(**kw).fit_transform(g)
    auc_single = link_auc(g, single[0], single[1])
    assert auc_single > 0.9
    for world in (1, 2, 3):
        def rank_fn(comm):
            m = E.models.SkipGram(**kw)
            c, x = m.fit_transform_blocks(g, comm, round_walks=300)
            return c[:, :16].cpu().numpy(), x[:, :16].cpu().numpy(), m.last_stats["pairs"]

        res = run_ranks(world, rank_fn)
        for r in res[1:]:
            assert np.array_equal(r[0], res[0][0]) and np.array_equal(r[1], res[0][1])
        assert np.isfinite(res[0][0]).all()
        assert link_auc(g, res[0][0], res[0][1]) > auc_single - 0.05
    with pytest.raises(NotImplementedError):
        E.models.CBOW(**kw).fit_transform_blocks(g, None)


def test_fused_pair_keys_group_and_shuffle(karate):
    """gn2v_walk_pair_blocks: sorting by the emitted key yields exactly the pair multiset of
    gn2v_walk_pairs, grouped by (centre % world, context % world), shuffled inside a block, with
    the unused slots (trap-node suffixes, trimmed windows) at the end."""
    wk = ops.walks(karate, ops.walk_params(24, 4, 0.5, 2.0), 5, 0, 0, 136)
    wk[::5, 9:] = -1
    world, window = 3, 4
    want = ops.walk_pairs(wk, window, 1).cpu().numpy().view(np.uint32)
    slots, keys = ops.walk_pair_blocks(wk, window, 1, world, 99)
    keys_s, order = torch.sort(keys)
    n = int((keys_s != 0x7FFFFFFFFFFFFFFF).sum())
    assert n == len(want)
    got = slots[order[:n]].cpu().numpy().view(np.uint32)
    block = (got[:, 0] % world).astype(np.int64) * world + got[:, 1] % world
    assert (np.diff(block) >= 0).all() and (keys_s[:n].cpu().numpy() >> 31 == block).all()
    as_set = lambda p: np.sort(p[:, 0].astype(np.int64) * 64 + p[:, 1])  # noqa: E731
    assert np.array_equal(as_set(got), as_set(want))
    inside = got[block == 4]
    assert not np.array_equal(inside, inside[np.lexsort((inside[:, 1], inside[:, 0]))])  # shuffled
    _, keys2 = ops.walk_pair_blocks(wk, window, 1, world, 100)
    assert not torch.equal(keys, keys2)  # the salt changes the shuffle


def test_one_rank_rccl_group_equals_loopback():
    """The trainer's collectives on the real backend ("nccl" = RCCL) with device tensors; a one-GPU
    box can only host a one-rank group, the 2-rank exchange logic is covered on gloo
    (tests/test_blocks_cpu.py)."""
    import os
    import subprocess
    import sys

    script = os.path.join(os.path.dirname(os.path.abspath(__file__)), "nccl_single_rank_check.py")
    res = subprocess.run([sys.executable, script], capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stderr[-3000:]
    ok = [l.split() for l in res.stdout.splitlines() if l.startswith("OK ")]  # RCCL logs to stdout too
    assert len(ok) == 1 and float(ok[0][1]) == 0.0, res.stdout[-500:]

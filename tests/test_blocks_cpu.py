"""Block-partitioned multi-GPU trainer (embiggen_amd/distributed.py), host logic on CPU with the
oracle's pair-mode step as compute stand-in (tests only): pair routing, block bucketing, ring
rotation of the context partitions and their mapping onto torch.distributed (2 gloo ranks)."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import embiggen_amd as E
from embiggen_amd.distributed import BlockPartitionedTrainer, LoopbackComm, TorchComm
from oracle import oracle as O
from sharded_helpers import host_init_fn, oracle_block_compute, run_ranks

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
D, K, W, L = 8, 4, 3, 14


def _train(comm, rounds=2, walks_per_round=9):
    g = E.karate_club()
    og = O.OracleGraph(g.row_ptr, g.col_idx)
    otp = O.TrainParams(0, D, D, 1, K, 1, 0.02, 0.9, 6.0, 1, D ** -0.5)
    tr = BlockPartitionedTrainer(g, otp, D, D, 42, D ** -0.5, comm, "cpu",
                                 init_fn=host_init_fn(34, D, D, 42, D ** -0.5))
    tr.compute = oracle_block_compute(og, otp, tr)
    wp = O.WalkParams(L, 1, 0.25, 4.0, 100, 0)
    for r in range(rounds):
        first = (r * comm.world + comm.rank) * walks_per_round
        pairs = O.walk_pairs(O.walks(og, wp, 42, 0, first, walks_per_round), W)
        tr.train_round(None, W, 1, 42, 0, 0.02, pairs=torch.from_numpy(pairs.view(np.int32)))
        assert tr.resident == comm.rank  # every partition is home again after a round
    return [t.numpy().copy() for t in tr.gather_full()], tr.last_round


def test_pair_extraction_counts():
    g = E.karate_club()
    og = O.OracleGraph(g.row_ptr, g.col_idx)
    walks = O.walks(og, O.WalkParams(L, 1, 1.0, 1.0, 100, 0), 1, 0, 0, 5)
    pairs = O.walk_pairs(walks, W)
    assert pairs.shape == (5 * (2 * W * L - W * (W + 1)), 2)
    assert pairs[0].tolist() == [walks[0, 0], walks[0, 1]]
    assert O.walk_pairs(walks, W, W).shape[0] == 5 * 2 * (L - W)
    cut = walks.copy()
    cut[0, 4:] = O.SENTINEL
    assert O.walk_pairs(cut, W).shape[0] == 4 * (2 * W * L - W * (W + 1)) + (2 * 3 * 4 - 3 * 4)


def _salt(n, seed, rnd):
    """The trainer's in-bucket shuffle key, restated with numpy."""
    idx = np.arange(n, dtype=np.int64)
    salt = (idx * 0x3C6EF35F + (seed * 0x19660D + rnd * 0x2545F491 + 1)) & 0x7FFFFFFF
    salt = ((salt ^ (salt >> 15)) * 0x2C1B3C6D) & 0x7FFFFFFF
    salt = ((salt ^ (salt >> 12)) * 0x297A2D39) & 0x7FFFFFFF
    return salt ^ (salt >> 15)


def test_world_one_is_the_shuffled_pair_list_trained_in_order():
    (c, x), info = _train(LoopbackComm())
    g = E.karate_club()
    og = O.OracleGraph(g.row_ptr, g.col_idx)
    otp = O.TrainParams(0, D, D, 1, K, 1, 0.02, 0.9, 6.0, 1, D ** -0.5)
    rc = O.init_table(34, D, D, 42, 0, D ** -0.5)
    rx = O.init_table(34, D, D, 42, 1, D ** -0.5)
    wp = O.WalkParams(L, 1, 0.25, 4.0, 100, 0)
    for r in range(2):
        pairs = O.walk_pairs(O.walks(og, wp, 42, 0, r * 9, 9), W)
        salt = _salt(len(pairs), 42, r)
        assert len(np.unique(salt)) > 0.99 * len(pairs)
        pairs = pairs[np.argsort(salt, kind="stable")]
        O.train_walks_ex(og, otp, pairs, 42, 0, r << 32, 0.02, rc, rx, neg_pool=g.col_idx,
                         neg_id_mul=1, neg_id_add=0, pair_mode=True)
    assert np.array_equal(c, rc) and np.array_equal(x, rx)
    assert info["pairs_generated"] == info["pairs_trained"] == 9 * (2 * W * L - W * (W + 1))


@pytest.mark.parametrize("world", [2, 3])
def test_blocks_cover_every_pair_once_and_rows_are_never_shared(world):
    sims = run_ranks(world, lambda comm: _train(comm, rounds=1))
    total = sum(s[1]["pairs_trained"] for s in sims)
    assert total == sum(s[1]["pairs_generated"] for s in sims)
    for r, ((c, x), info) in enumerate(sims):
        assert sum(info["block_sizes"]) == info["pairs_trained"]
    # all ranks assemble identical full tables
    for s in sims[1:]:
        assert np.array_equal(s[0][0], sims[0][0][0]) and np.array_equal(s[0][1], sims[0][0][1])


def _gloo_worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    (c, x), _ = _train(TorchComm())
    np.save(os.path.join(out_dir, f"c{rank}.npy"), c)
    np.save(os.path.join(out_dir, f"x{rank}.npy"), x)
    dist.barrier()
    dist.destroy_process_group()


def _free_port():
    import socket

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.parametrize("world", [2, 3])
def test_gloo_ranks_equal_the_in_process_simulation(tmp_path, world):
    """No row is ever shared between ranks, so the distributed run is exactly the simulation
    (world 3: ragged partitions of 12 / 11 / 11 rows go round the ring)."""
    mp.spawn(_gloo_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    sim = run_ranks(world, lambda comm: _train(comm))
    for r in range(world):
        assert np.array_equal(np.load(tmp_path / f"c{r}.npy"), sim[r][0][0])
        assert np.array_equal(np.load(tmp_path / f"x{r}.npy"), sim[r][0][1])
    init = O.init_table(34, D, D, 42, 0, D ** -0.5)
    assert np.abs(sim[0][0][0] - init).max() > 1e-3

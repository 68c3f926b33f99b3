"""Row-sharded trainer (embiggen_amd/distributed.py), host logic on CPU.  The fused kernel is
replaced by the oracle's general step (tests only); under test are ownership, the request /
gather / delta-scatter exchange and its mapping onto torch.distributed (2 gloo ranks)."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import embiggen_amd as E
from embiggen_amd.distributed import LoopbackComm, RowShardedTables, ShardedTrainer, TorchComm
from oracle import oracle as O
from sharded_helpers import host_init_fn, oracle_compute, run_ranks

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
D, K, W, L = 8, 4, 3, 14


def _params(model):
    return O.TrainParams(model, D, D, 1, K, W, 0.02, 0.9, 6.0, 1, D ** -0.5)


def _train(comm, model, n_batches=3, walks_per_batch=17):
    g = E.karate_club()
    og = O.OracleGraph(g.row_ptr, g.col_idx)
    tables = RowShardedTables(g, D, D, 42, D ** -0.5, comm, "cpu",
                              init_fn=host_init_fn(34, D, D, 42, D ** -0.5))
    otp = _params(model)
    trainer = ShardedTrainer(g, tables, otp, compute=oracle_compute(og, otp, tables))
    wp = O.WalkParams(L, 1, 0.25, 4.0, 100, 0)
    for b in range(n_batches):
        first = (b * comm.world + comm.rank) * walks_per_batch
        walks = torch.from_numpy(O.walks(og, wp, 42, 0, first, walks_per_batch).view(np.int32))
        trainer.train_batch(walks, 42, 0, first, 0.02)
    full = tables.gather_full()
    return [t.numpy().copy() for t in full], trainer.last_exchange


@pytest.mark.parametrize("model", [0, 1])
def test_world_one_follows_the_cache_and_delta_protocol(model):
    """world = 1 restated directly: per batch the walk nodes' rows are copied into caches, the
    step runs on (caches, negatives in the table itself), then table += cache_new - cache_old."""
    (c, x), info = _train(LoopbackComm(), model)
    g = E.karate_club()
    og = O.OracleGraph(g.row_ptr, g.col_idx)
    rc = O.init_table(34, D, D, 42, 0, D ** -0.5)
    rx = O.init_table(34, D, D, 42, 1, D ** -0.5)
    wp = O.WalkParams(L, 1, 0.25, 4.0, 100, 0)
    for b in range(3):
        walks = O.walks(og, wp, 42, 0, b * 17, 17)
        old_c, old_x = rc.copy(), rx.copy()
        cache_c, cache_x = rc.copy(), rx.copy()
        O.train_walks_ex(og, _params(model), walks, 42, 0, b * 17, 0.02, cache_c, cache_x,
                         walk_rows=walks, negative=rc if model == 1 else rx, neg_pool=g.col_idx,
                         neg_id_mul=1, neg_id_add=0)
        rc += cache_c - old_c
        rx += cache_x - old_x
    assert np.allclose(c, rc, atol=1e-6) and np.allclose(x, rx, atol=1e-6)
    assert info["unique_nodes"] <= 34 and info["rows_sent"] == info["unique_nodes"]


@pytest.mark.parametrize("model", [0, 1])
def test_ownership_and_pools(model):
    results = run_ranks(3, lambda comm: RowShardedTables(
        E.karate_club(), D, D, 42, D ** -0.5, comm, "cpu",
        init_fn=host_init_fn(34, D, D, 42, D ** -0.5)))
    g = E.karate_club()
    full = O.init_table(34, D, D, 42, 0, D ** -0.5)
    assert sum(t.n_local for t in results) == 34
    for r, t in enumerate(results):
        assert np.array_equal(t.central.numpy(), full[r::3])
        pool_global = t.neg_pool.numpy().astype(np.int64) * 3 + r
        assert np.array_equal(np.sort(pool_global), np.sort(g.col_idx[g.col_idx % 3 == r]))


def _gloo_worker(rank, world, port, model, out_dir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    (c, x), _ = _train(TorchComm(), model)
    np.save(os.path.join(out_dir, f"c{rank}.npy"), c)
    np.save(os.path.join(out_dir, f"x{rank}.npy"), x)
    dist.barrier()
    dist.destroy_process_group()


def _free_port():
    import socket

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.parametrize("model", [0, 1])
def test_two_gloo_ranks_equal_the_in_process_simulation(tmp_path, model):
    """The torch.distributed mapping (all_to_all_single with uneven splits) moves exactly what the
    in-process reference communicator moves; replicas of the gathered tables agree."""
    world = 2
    mp.spawn(_gloo_worker, args=(world, _free_port(), model, str(tmp_path)), nprocs=world,
             join=True)
    sim = run_ranks(world, lambda comm: _train(comm, model))
    for r in range(world):
        c, x = np.load(tmp_path / f"c{r}.npy"), np.load(tmp_path / f"x{r}.npy")
        assert np.allclose(c, sim[r][0][0], atol=1e-6) and np.allclose(x, sim[r][0][1], atol=1e-6)
    assert np.array_equal(np.load(tmp_path / "c0.npy"), np.load(tmp_path / "c1.npy"))
    # training happened, and differs from the world = 1 trajectory only through shard-local
    # negatives / batch-level staleness (same quality is checked on the GPU at scale)
    init = O.init_table(34, D, D, 42, 0, D ** -0.5)
    assert np.abs(sim[0][0][0] - init).max() > 1e-3


def test_deltas_of_rows_shared_by_several_ranks_are_summed():
    """A node visited by walks of both ranks receives both ranks' updates."""
    sim2 = run_ranks(2, lambda comm: _train(comm, 0, n_batches=1))
    (c2, x2), info = sim2[0]
    assert info["rows_served"] > 0
    g = E.karate_club()
    og = O.OracleGraph(g.row_ptr, g.col_idx)
    wp = O.WalkParams(L, 1, 0.25, 4.0, 100, 0)
    visited = [set(O.walks(og, wp, 42, 0, r * 17, 17).ravel().tolist()) for r in range(2)]
    shared = sorted(visited[0] & visited[1])
    assert shared  # hubs 0 and 33 are on almost every walk
    init = O.init_table(34, D, D, 42, 0, D ** -0.5)
    # central rows move only for visited nodes; shared ones moved by the sum of two deltas
    untouched = sorted(set(range(34)) - visited[0] - visited[1])
    assert np.array_equal(c2[untouched], init[untouched])
    assert (np.abs(c2[shared] - init[shared]).max(1) > 0).all()

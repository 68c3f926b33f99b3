"""N > 1 host logic on CPU: two gloo ranks.  The compute stand-in is the oracle (allowed in tests
only); what is under test is the product's partitioning + replica exchange
(embiggen_amd/distributed.py), i.e. exactly what bench.py --gpus N runs around the HIP kernels."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import embiggen_amd as E
    from embiggen_amd.distributed import ReplicaSync, walk_slice
    from oracle import oracle as O

    g = E.karate_club()
    og = O.OracleGraph(g.row_ptr, g.col_idx)
    d = 8
    wp = O.WalkParams(12, 1, 0.25, 4.0, 100, 0)
    tp = O.TrainParams(0, d, d, 1, 4, 2, 0.02, 0.9, 6.0, 1, d ** -0.5)
    central = torch.from_numpy(O.init_table(34, d, d, 42, 0, d ** -0.5))
    contextual = torch.from_numpy(O.init_table(34, d, d, 42, 1, d ** -0.5))
    sync = ReplicaSync(central, contextual)
    for step in range(3):
        first, count = walk_slice(step, rank, world, 17)
        w = O.walks(og, wp, 42, 0, first, count)
        O.train_walks(og, tp, w, 42, 0, first, 0.02, central.numpy(), contextual.numpy())
        sync.sync()
    np.save(os.path.join(out_dir, f"central_{rank}.npy"), central.numpy())
    np.save(os.path.join(out_dir, f"contextual_{rank}.npy"), contextual.numpy())
    dist.barrier()
    dist.destroy_process_group()


def _free_port():
    import socket

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_walk_slices_are_disjoint_and_contiguous():
    from embiggen_amd.distributed import walk_slice

    seen = []
    for step in range(4):
        for rank in range(3):
            first, n = walk_slice(step, rank, 3, 10)
            seen.extend(range(first, first + n))
    assert seen == list(range(4 * 3 * 10))


def test_two_rank_delta_sum_equals_sequential_delta_application(tmp_path):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    c = [np.load(tmp_path / f"central_{r}.npy") for r in range(world)]
    x = [np.load(tmp_path / f"contextual_{r}.npy") for r in range(world)]
    assert np.array_equal(c[0], c[1]) and np.array_equal(x[0], x[1])  # replicas agree

    # single-process restatement: per step, every rank's slice is trained from the same base and
    # the deltas are summed
    sys.path.insert(0, ROOT)
    import embiggen_amd as E
    from embiggen_amd.distributed import walk_slice
    from oracle import oracle as O

    g = E.karate_club()
    og = O.OracleGraph(g.row_ptr, g.col_idx)
    d = 8
    wp = O.WalkParams(12, 1, 0.25, 4.0, 100, 0)
    tp = O.TrainParams(0, d, d, 1, 4, 2, 0.02, 0.9, 6.0, 1, d ** -0.5)
    base_c = O.init_table(34, d, d, 42, 0, d ** -0.5)
    base_x = O.init_table(34, d, d, 42, 1, d ** -0.5)
    for step in range(3):
        dc, dx = np.zeros_like(base_c), np.zeros_like(base_x)
        for rank in range(world):
            cc, xx = base_c.copy(), base_x.copy()
            first, count = walk_slice(step, rank, world, 17)
            O.train_walks(og, tp, O.walks(og, wp, 42, 0, first, count), 42, 0, first, 0.02, cc, xx)
            dc += cc - base_c
            dx += xx - base_x
        base_c, base_x = base_c + dc, base_x + dx
    assert np.allclose(c[0], base_c, atol=1e-6) and np.allclose(x[0], base_x, atol=1e-6)
    assert np.abs(c[0] - O.init_table(34, d, d, 42, 0, d ** -0.5)).max() > 1e-3


def test_single_process_sync_is_a_no_op():
    from embiggen_amd.distributed import ReplicaSync

    t = torch.ones(4, 4)
    ReplicaSync(t).sync()
    assert torch.equal(t, torch.ones(4, 4))

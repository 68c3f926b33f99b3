"""GPU: the row-sharded trainer with the real fused kernel (gn2v_step) on row caches.
Several ranks are simulated on the one GPU of the test box by threads sharing an in-process
communicator; the multi-GPU RCCL run itself is the driver's."""
import numpy as np
import pytest
import torch

import embiggen_amd as E
from embiggen_amd import _lib, ops
from embiggen_amd.distributed import RowShardedTables, ShardedTrainer
from oracle import oracle as O
from sharded_helpers import host_init_fn, oracle_compute, run_ranks

pytestmark = pytest.mark.gpu
D, K, W, L = 16, 4, 3, 14


def _run(comm, model, device, use_oracle):
    g = E.karate_club()
    og = O.OracleGraph(g.row_ptr, g.col_idx)
    if device == "cpu":
        tables = RowShardedTables(g, D, D, 42, D ** -0.5, comm, "cpu",
                                  init_fn=host_init_fn(34, D, D, 42, D ** -0.5))
    else:
        tables = RowShardedTables(g, D, D, 42, D ** -0.5, comm, "cuda:0")
    otp = O.TrainParams(model, D, D, 1, K, W, 0.02, 0.9, 6.0, 1, D ** -0.5)
    tp = ops.train_params(model, D, K, W, flags=1 | _lib.TRAIN_DETERMINISTIC)
    trainer = ShardedTrainer(g, tables, tp if not use_oracle else otp,
                             compute=oracle_compute(og, otp, tables) if use_oracle else None)
    wp = O.WalkParams(L, 1, 0.25, 4.0, 100, 0)
    for b in range(3):
        first = (b * comm.world + comm.rank) * 17
        walks = torch.from_numpy(O.walks(og, wp, 42, 0, first, 17).view(np.int32)).to(device)
        trainer.train_batch(walks, 42, 0, first, 0.02)
    return [t.cpu().numpy() for t in tables.gather_full()]


@pytest.mark.parametrize("world", [1, 2, 3])
@pytest.mark.parametrize("model", [0, 1])
def test_sharded_trainer_kernel_equals_oracle(world, model):
    gpu = run_ranks(world, lambda comm: _run(comm, model, "cuda:0", use_oracle=False))
    ref = run_ranks(world, lambda comm: _run(comm, model, "cpu", use_oracle=True))
    for r in range(world):
        assert np.abs(gpu[r][0] - ref[r][0]).max() < 1e-5
        assert np.abs(gpu[r][1] - ref[r][1]).max() < 1e-5
    assert np.array_equal(gpu[0][0], gpu[-1][0])  # every rank gathers the same full table


def _auc(g, c, x, gen, n_eval=100000):
    t = g._device_tensors
    n = g.get_number_of_nodes()
    e = torch.randint(0, t["col_idx"].numel(), (n_eval,), device="cuda", generator=gen)
    dst = t["col_idx"][e].long()
    src = torch.searchsorted(t["row_ptr"], e, right=True) - 1
    ru = torch.randint(0, n, (n_eval,), device="cuda", generator=gen)
    rv = torch.randint(0, n, (n_eval,), device="cuda", generator=gen)
    score = lambda u, v: (c[u] * x[v]).sum(1) + (c[v] * x[u]).sum(1)  # noqa: E731
    s = torch.cat([score(src, dst), score(ru, rv)])
    ranks = torch.empty_like(s)
    ranks[torch.argsort(s)] = torch.arange(1, s.numel() + 1, device="cuda", dtype=s.dtype)
    return float((ranks[:n_eval].sum() - n_eval * (n_eval + 1) / 2) / (n_eval * n_eval))


def test_sharded_training_reaches_single_gpu_quality_at_scale():
    """BA 200 k nodes, d = 64: two simulated ranks (shard-local negatives, per-batch exchange)
    reach the link-prediction AUROC of the unsharded trainer on the same walk budget."""
    g = E.barabasi_albert(200_000, 8, 42)
    n, d = g.get_number_of_nodes(), 64
    wp = ops.walk_params(64, 1, 1.0, 1.0)
    tp = ops.train_params(0, d, 5, 4, flags=1)
    total, batch = 1 << 16, 1 << 13

    c = ops.init_table(n, d, 42, 0, d ** -0.5)
    x = ops.init_table(n, d, 42, 1, d ** -0.5)
    for first in range(0, total, batch):
        ops.sgns_step(g, tp, ops.walks(g, wp, 42, 0, first, batch), 42, 0, first, 0.025, c, x)
    gen = torch.Generator(device="cuda")
    gen.manual_seed(1)
    auc_single = _auc(g, c, x, gen)

    def rank_fn(comm):
        tables = RowShardedTables(g, d, d, 42, d ** -0.5, comm, "cuda:0")
        trainer = ShardedTrainer(g, tables, tp)
        for step in range(total // batch // comm.world):
            first = (step * comm.world + comm.rank) * batch
            trainer.train_batch(ops.walks(g, wp, 42, 0, first, batch), 42, 0, first, 0.025)
        return tables.gather_full(), trainer.last_exchange

    (sc, sx), info = run_ranks(2, rank_fn)[0]
    gen.manual_seed(1)
    auc_sharded = _auc(g, sc, sx, gen)
    assert bool(torch.isfinite(sc).all()) and info["unique_nodes"] > 10000
    assert auc_single > 0.6 and auc_sharded > auc_single - 0.03, (auc_sharded, auc_single)

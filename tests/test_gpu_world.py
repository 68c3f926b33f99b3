"""``gn2v_train_world`` (include/gn2v.h): the multi-GPU fit driven from C through a communicator the
host fills -- the form of ``self._model.fit_transform(graph)`` (embedders/ensmallen_embedders/
node2vec.py:99) a non-Python binding calls on every rank.  Here the ranks are threads of one
process that share GPU 0 (tests/sharded_helpers.py ThreadComm, wrapped as a ``gn2v_comm`` of
callbacks by ``embiggen_amd.distributed.CComm``); each rank has a graph handle of its own.  The
deterministic kernels make the comparison exact: the C loop must produce the tables of the Python
trainer (``BlockPartitionedTrainer``: the schedule the gloo tests of tests/test_blocks_cpu.py pin
on the oracle) bit for bit, on every rank."""
import numpy as np
import pytest
import torch

import embiggen_amd as E
from embiggen_amd import models
from embiggen_amd.distributed import LoopbackComm
from oracle import oracle as O
from sharded_helpers import run_ranks

pytestmark = pytest.mark.gpu


def _graph(nodes, m, seed=9):
    s, d = O.ba_edges(nodes, m, seed)
    return E.CSRGraph.from_edge_list(s, d, number_of_nodes=nodes)  # a handle of its own per call


def _fit(comm, loop, nodes, m, kw, round_walks, budget):
    g = _graph(nodes, m)
    model = models.SkipGram(**kw)
    model.deterministic = True
    if loop == "c":
        c, x = model.fit_transform_world(g, comm, round_walks=round_walks,
                                         max_walks_per_epoch=budget)
    else:
        c, x = model.fit_transform_blocks(g, comm, round_walks=round_walks, overlap=False,
                                          max_walks_per_epoch=budget)
    torch.cuda.synchronize()
    return c.cpu().numpy(), x.cpu().numpy(), dict(model.last_stats), dict(model.last_plan)


@pytest.mark.timeout(900)
@pytest.mark.parametrize("world,nodes,m,d,budget,round_walks", [
    (2, 3_000, 4, 16, 0, 700),        # XCD cells, hot rows, four parts, ragged last round
    (3, 5_000, 3, 24, 0, 1_000),      # three ranks: six parts
    (2, 120_000, 5, 128, 6_000, 1_500),  # resident cells under a placement per round
])
def test_c_world_loop_equals_the_python_trainer(world, nodes, m, d, budget, round_walks):
    kw = dict(embedding_size=d, epochs=2, walk_length=16, iterations=1, window_size=3,
              number_of_negative_samples=4, learning_rate=0.05, random_state=7, verbose=False)
    got = run_ranks(world, lambda comm: _fit(comm, "c", nodes, m, kw, round_walks, budget))
    want = run_ranks(world, lambda comm: _fit(comm, "python", nodes, m, kw, round_walks, budget))
    walks = budget or nodes
    pairs = sum(r[2]["pairs"] for r in got)
    assert pairs == sum(r[2]["pairs"] for r in want) > 0
    assert pairs == 2 * walks * (2 * 3 * 16 - 3 * 4)  # every pair of every walk, once, on some rank
    for rank in range(world):
        assert got[rank][3]["parts"] == want[rank][3]["parts"]
        assert got[rank][3]["slices"] == want[rank][3]["slices"]
        for t in (0, 1):
            assert np.isfinite(got[rank][t]).all()
            assert np.array_equal(got[rank][t], want[rank][t]), (rank, t, np.abs(
                got[rank][t] - want[rank][t]).max())
            assert np.array_equal(got[rank][t], got[0][t])  # every rank receives the same tables
    if nodes >= 100_000:
        assert got[0][3]["slices"] > 16 and got[0][2]["resident_launches"] > 0
    init = O.init_table(nodes, d, (d + 31) // 32 * 32 if d > 16 else d, 7, 0, d ** -0.5)
    assert np.abs(got[0][0][:, :d] - init[:, :d]).max() > 1e-3  # and they moved


def test_c_world_loop_with_one_rank_and_a_copying_communicator():
    """world = 1 through the same code (the parts are separate buffers, the placement keeps the
    classes modulo `parts`): every pair trained once, finite tables that moved; and a failing
    callback comes back as the Python exception it was."""
    nodes, d = 120_000, 32
    kw = dict(embedding_size=d, epochs=1, walk_length=16, iterations=1, window_size=3,
              number_of_negative_samples=4, random_state=7, verbose=False)
    g = _graph(nodes, 5)
    model = models.SkipGram(**kw)
    c, x = model.fit_transform_world(g, LoopbackComm(), max_walks_per_epoch=20_000)
    assert model.last_stats["pairs"] == 20_000 * (2 * 3 * 16 - 3 * 4)
    assert model.last_stats["resident_launches"] > 0 and model.last_plan["slices"] > 16
    assert bool(torch.isfinite(c).all()) and bool(torch.isfinite(x).all())

    class Broken(LoopbackComm):
        def all_gather(self, tensor):
            raise RuntimeError("the fabric is down")

    with pytest.raises(RuntimeError, match="the fabric is down"):
        models.SkipGram(**kw).fit_transform_world(g, Broken(), max_walks_per_epoch=2_000)


@pytest.mark.timeout(1500)
def test_config4_at_full_size_with_eight_ranks_through_the_c_loop():
    """BASELINE config 4 ("embedding table row-sharded across 8 x MI355X": 2 449 029 nodes /
    61 M edges, d = 128) with eight rank-threads on one GPU, every rank with a graph handle of its
    own, their rounds driven by ``gn2v_train_world`` in its shipping (parallel) form: 16
    travelling parts of resident cells under a placement per round.  Every pair of every walk is
    trained exactly once on some rank, every rank receives the same full tables, they are finite
    and separate edges from random pairs."""
    from sharded_helpers import link_auc_device

    nodes, world, budget = 2_449_029, 8, 1 << 21
    kw = dict(embedding_size=128, epochs=1, walk_length=128, iterations=10, window_size=5,
              number_of_negative_samples=10, learning_rate=0.025, random_state=42, verbose=False)

    def rank_fn(comm):
        g = E.barabasi_albert(nodes, 25, 42)  # a handle of its own
        model = models.SkipGram(**kw)
        c, x = model.fit_transform_world(g, comm, max_walks_per_epoch=budget)
        torch.cuda.synchronize()
        out = (dict(model.last_stats), dict(model.last_plan),
               bool(torch.isfinite(c).all() and torch.isfinite(x).all()),
               int(c.view(torch.int32).sum(dtype=torch.int64)),
               int(x.view(torch.int32).sum(dtype=torch.int64)))
        if comm.rank == 0:
            gen = torch.Generator(device="cuda")
            gen.manual_seed(1)
            out += (link_auc_device(g, c, x, gen),)
        return out

    res = run_ranks(world, rank_fn)
    assert sum(r[0]["pairs"] for r in res) == budget * (2 * 5 * 128 - 5 * 6)
    assert all(r[1]["world"] == 8 and r[1]["parts"] == 16 and r[1]["slices"] > 16 for r in res)
    assert all(r[0]["resident_launches"] > 0 and r[2] for r in res)
    assert len({(r[3], r[4]) for r in res}) == 1  # bit-identical tables on every rank
    assert res[0][5] > 0.6, res[0][5]  # one walk per node (the simulated-rank test of config 4: 0.6+)

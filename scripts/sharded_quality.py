"""Quality of row-sharded training (simulated ranks on one GPU) vs the unsharded trainer."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import embiggen_amd as E
from embiggen_amd import ops
from embiggen_amd.distributed import RowShardedTables, ShardedTrainer
from sharded_helpers import run_ranks
from test_gpu_sharded import _auc

nodes, m, d = int(sys.argv[1]), 8, 64
total, batch, world = int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
merge = sys.argv[5] if len(sys.argv) > 5 else "mean"
g = E.barabasi_albert(nodes, m, 42)
n = g.get_number_of_nodes()
wp = ops.walk_params(64, 1, 1.0, 1.0)
tp = ops.train_params(0, d, 5, 4, flags=1)
gen = torch.Generator(device="cuda")
c = ops.init_table(n, d, 42, 0, d ** -0.5); x = ops.init_table(n, d, 42, 1, d ** -0.5)
t0 = time.time()
for first in range(0, total, batch):
    ops.sgns_step(g, tp, ops.walks(g, wp, 42, 0, first, batch), 42, 0, first, 0.025, c, x)
torch.cuda.synchronize(); t1 = time.time()
gen.manual_seed(1); print("single", _auc(g, c, x, gen), f"{t1 - t0:.2f}s", flush=True)
def rank_fn(comm):
    tables = RowShardedTables(g, d, d, 42, d ** -0.5, comm, "cuda:0")
    trainer = ShardedTrainer(g, tables, tp, merge=merge)
    for step in range(total // batch // comm.world):
        first = (step * comm.world + comm.rank) * batch
        trainer.train_batch(ops.walks(g, wp, 42, 0, first, batch), 42, 0, first, 0.025)
    return tables.gather_full(), trainer.last_exchange
t0 = time.time()
(sc, sx), info = run_ranks(world, rank_fn)[0]
torch.cuda.synchronize(); t1 = time.time()
gen.manual_seed(1); print("sharded", merge, world, _auc(g, sc, sx, gen), info, f"{t1 - t0:.2f}s", flush=True)

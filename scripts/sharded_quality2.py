import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import embiggen_amd as E
from embiggen_amd import ops, _lib
from embiggen_amd.distributed import RowShardedTables, ShardedTrainer, LoopbackComm
from test_gpu_sharded import _auc
nodes, total = int(sys.argv[1]), int(sys.argv[2])
lr = float(sys.argv[3]) if len(sys.argv) > 3 else 0.025
g = E.barabasi_albert(nodes, 8, 42); n = g.get_number_of_nodes(); d = 64
wp = ops.walk_params(64, 1, 1.0, 1.0)
gen = torch.Generator(device="cuda")
for mode_flag, label in ((_lib.TRAIN_ATOMIC, "atomic"), (_lib.TRAIN_WRITE_THROUGH, "wt")):
    tp = ops.train_params(0, d, 5, 4, flags=1 | mode_flag)
    for batch in (1024, 8192, 65536):
        c = ops.init_table(n, d, 42, 0, d ** -0.5); x = ops.init_table(n, d, 42, 1, d ** -0.5)
        for first in range(0, total, batch):
            ops.sgns_step(g, tp, ops.walks(g, wp, 42, 0, first, batch), 42, 0, first, lr, c, x)
        gen.manual_seed(1); a1 = _auc(g, c, x, gen)
        tables = RowShardedTables(g, d, d, 42, d ** -0.5, LoopbackComm(), "cuda:0")
        tr = ShardedTrainer(g, tables, tp)
        for first in range(0, total, batch):
            tr.train_batch(ops.walks(g, wp, 42, 0, first, batch), 42, 0, first, lr)
        sc, sx = tables.gather_full()
        gen.manual_seed(1); a2 = _auc(g, sc, sx, gen)
        print(f"{label} batch {batch}: single {a1:.4f}  sharded(world=1) {a2:.4f}", flush=True)

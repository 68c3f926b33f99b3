"""Wall time of one block-trainer round against the time its training kernels take (one GPU)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
import embiggen_amd as E
from embiggen_amd import _lib, ops
from embiggen_amd.distributed import BlockPartitionedTrainer, LoopbackComm
g = E.barabasi_albert(10_000_000, 10, 42); d = 128
nw = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
tp = ops.train_params(0, d, 10, 1, flags=1)
tr = BlockPartitionedTrainer(g, tp, d, 128, 42, d ** -0.5, LoopbackComm(), "cuda:0")
wp = ops.walk_params(128, 10, 0.25, 4.0)
for rep in range(3):
    ops.stats_reset(g)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    wk = ops.walks(g, wp, 42, 0, rep * nw, nw)
    tr.train_round(wk, 5, 1, 42, 0, 0.01)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    st = ops.stats_read(g)
    print(f"round {dt * 1e3:.0f} ms, training kernels {st['train_ms']:.0f} ms, other {dt * 1e3 - st['train_ms']:.0f} ms, "
          f"{st['pairs'] / dt:.3e} pairs/s, peak {torch.cuda.max_memory_allocated() / 1e9:.0f} GB", flush=True)

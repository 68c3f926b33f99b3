"""Where does the block kernel's time go?  Time of one round (2^19 walks, every part) on the bench
graph as a function of the number of negatives and of the record length."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

import embiggen_amd as E
from embiggen_amd import ops
from embiggen_amd.distributed import BlockPartitionedTrainer, LoopbackComm

g = E.barabasi_albert(10_000_000, 10, 42)
d = 128
wp = ops.walk_params(128, 10, 0.25, 4.0)
from embiggen_amd import _lib

for k, record, extra, tag in ((10, 16, 0, "default"), (2, 16, 0, "default"), (0, 16, 0, "default"),
                              (10, 32, 0, "default"), (22, 16, 0, "default"),
                              (0, 16, _lib.TRAIN_WRITE_BACK, "central rows by plain stores"),
                              (10, 16, _lib.TRAIN_WRITE_BACK, "central rows by plain stores")):
    tp = ops.train_params(0, d, k, 5, flags=1 | extra, ld=d)
    tr = BlockPartitionedTrainer(g, tp, d, d, 42, d ** -0.5, LoopbackComm(), "cuda:0",
                                 walk_length=128, window=5, record=record)
    wk = ops.walks(g, wp, 42, 0, 0, 1 << 19)
    # (resident cells: the round's placement, its alias tables and the placed walks)
    prepared = tr.prepare(wk, 42, 0, 0, rstate=tr.round_state(wk, 42, 0))
    tr.train_prepared(prepared, 42, 0, 0.01)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    tr.train_prepared(prepared, 42, 0, 0.01)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) * 1e3
    pairs = prepared[2]
    rows = pairs * (k + 1) + pairs / record
    print(f"k={k:2d} record={record} {tag}: {ms:8.1f} ms  {ms * 1e6 / pairs:6.2f} ns/pair  "
          f"{rows * 1024 / ms / 1e9:6.2f} TB/s of row traffic", flush=True)
    del tr, prepared

"""Node2VecSequence (the `embiggen.sequences` side of the path) measured: batches of the
reference's default shape (256 sources x 16 iterations, walks of 128, window 4: 491 520 rows of
8 + 1 int32 = 17.7 MB) and of a large batch, on a BA graph; next to the CPU oracle's walks +
windows on the host's threads."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("OMP_NUM_THREADS", "16")  # the CPU leg: the box's usable cores, not its 256 logical ones
import numpy as np
import torch

import embiggen_amd as E
from oracle import oracle as O

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
g = E.barabasi_albert(n, 10, 42)


def timed(seq, batches, what):
    seq[0]
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    rows = 0
    for i in range(1, batches + 1):
        ((contexts, words),), = seq[i]
        rows += len(words)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    steps = batches * seq.batch_size * seq._iterations * (seq._walk_length - 1)
    print(f"{what}: {batches} batches in {dt:.3f} s = {dt / batches * 1e3:.2f} ms a batch, "
          f"{rows / dt:.3e} rows/s ({rows / dt * 36 / 1e9:.2f} GB/s of int32 output), "
          f"{steps / dt:.3e} walk steps/s", flush=True)


for kw, batches, what in (
        (dict(), 200, "default batch (256 x 16 walks), numpy out"),
        (dict(return_device_tensors=True), 200, "default batch, device tensors out"),
        (dict(batch_size=65536, return_device_tensors=True), 10, "batch_size 65 536, device tensors out"),
        (dict(batch_size=65536), 5, "batch_size 65 536, numpy out"),
        (dict(return_weight=0.25, explore_weight=4.0, batch_size=65536, return_device_tensors=True), 10,
         "batch_size 65 536, return 0.25 / explore 4, device tensors out")):
    timed(E.Node2VecSequence(g, **kw), batches, what)

# the CPU oracle on the same default batch (OpenMP over walks)
og = O.OracleGraph(g.row_ptr, g.col_idx)
wp = O.WalkParams(128, 16, 1.0, 1.0, 100, 0)
t0 = time.perf_counter()
reps = 5
for r in range(reps):
    wk = np.concatenate([O.walks(og, wp, 42 + r, 0, 2 * it * n, 256) for it in range(16)])
    O.window_batch(wk, 4)
dt = (time.perf_counter() - t0) / reps
print(f"CPU oracle, default batch: {dt * 1e3:.1f} ms a batch ({os.environ['OMP_NUM_THREADS']} threads), "
      f"{4096 * 127 / dt:.3e} walk steps/s")

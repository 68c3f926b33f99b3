"""Where does the CBOW kernel's time go?  Launch time of one 2^16-walk batch on the bench graph as
a function of the number of negatives (output rows per centre) and of the window (contextual rows
per centre; the LDS window cache)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

import embiggen_amd as E
from embiggen_amd import _lib, ops

g = E.barabasi_albert(10_000_000, 10, 42)
n, d = g.get_number_of_nodes(), 128
c = ops.init_table(n, d, 42, 0, d ** -0.5)
x = ops.init_table(n, d, 42, 1, d ** -0.5)
wk = ops.walks(g, ops.walk_params(128, 10, 0.25, 4.0), 42, 0, 0, 1 << 16)
for model, name in ((1, "cbow"), (0, "sgns")):
    for k, w in ((10, 5), (2, 5), (0, 5), (10, 1), (10, 2), (2, 1), (22, 5)):
        tp = ops.train_params(model, d, k, w, flags=1)
        step = ops.cbow_step if model else ops.sgns_step
        step(g, tp, wk, 42, 0, 0, 0.01, c, x)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for r in range(3):
            step(g, tp, wk, 42, 0, r << 20, 0.01, c, x)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / 3 * 1e3
        centres = (1 << 16) * 128
        print(f"{name} k={k:2d} window={w}: {ms:7.2f} ms  {ms * 1e6 / centres:6.2f} ns/centre", flush=True)

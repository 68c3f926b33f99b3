"""Small driver for rocprofv3 --pmc experiments on the SGNS kernel:
    rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d out -o x -- python3 scripts/pmc_probe.py --k 10 --window 5 --mode write_through
Prints pairs / centres per launch so counter values can be normalised."""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

import embiggen_amd as E  # noqa: E402
from embiggen_amd import _lib, ops  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--nodes", type=int, default=10_000_000)
ap.add_argument("--k", type=int, default=10)
ap.add_argument("--window", type=int, default=5)
ap.add_argument("--d", type=int, default=128)
ap.add_argument("--walks", type=int, default=1 << 15)
ap.add_argument("--reps", type=int, default=3)
ap.add_argument("--mode", default="write_through")
ap.add_argument("--model", type=int, default=0)
ap.add_argument("--ld", type=int, default=0)
a = ap.parse_args()
g = E.barabasi_albert(a.nodes, 10, 42)
n = g.get_number_of_nodes()
ld = a.ld or (a.d + 3) // 4 * 4
c = ops.init_table(n, a.d, 42, 0, a.d ** -0.5, ld=ld)
x = ops.init_table(n, a.d, 42, 1, a.d ** -0.5, ld=ld)
flags = 1 | {"write_through": _lib.TRAIN_WRITE_THROUGH, "write_back": _lib.TRAIN_WRITE_BACK,
             "atomic": _lib.TRAIN_ATOMIC}[a.mode]
tp = ops.train_params(a.model, a.d, a.k, a.window, flags=flags, ld=ld)
wk = ops.walks(g, ops.walk_params(128, 10, 0.25, 4.0), 42, 0, 0, a.walks)
step = ops.sgns_step if a.model == 0 else ops.cbow_step
for r in range(a.reps):
    ops.stats_reset(g)
    step(g, tp, wk, 42, 0, 0, 0.01, c, x)
    st = ops.stats_read(g)
print(json.dumps({"k": a.k, "window": a.window, "mode": a.mode, "model": a.model,
                  "pairs_per_launch": st["pairs"], "centres_per_launch": st["centres"],
                  "train_ms": st["train_ms"], "d": a.d, "ld": ld}))

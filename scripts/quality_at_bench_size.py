"""The quality gate's comparison (tests/test_gpu_quality_gates.py compare()) at the bench size with
enough training for the cosines to mean something: BA 10 M / 100 M, d = 128, reference defaults,
`--epochs` epochs of one walk a node (default 4: 5 x 10^10 pairs) -- the default schedule against
the walk-ordered one with graph-wide negatives on the same walks.  A log, not a test (minutes).
    python scripts/quality_at_bench_size.py [--epochs 4] [--nodes 10000000]
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import embiggen_amd as E  # noqa: E402
from test_gpu_quality_gates import compare  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--nodes", type=int, default=10_000_000)
ap.add_argument("--m", type=int, default=10)
ap.add_argument("--epochs", type=int, default=4)
ap.add_argument("--iterations", type=int, default=1)
a = ap.parse_args()
g = E.barabasi_albert(a.nodes, a.m, 42)
kw = dict(embedding_size=128, epochs=a.epochs, walk_length=128, iterations=a.iterations,
          window_size=5, number_of_negative_samples=10, return_weight=0.25, explore_weight=4.0,
          learning_rate=0.01, random_state=42, verbose=False)
res = compare(g, kw, "auto")
print(f"BA {a.nodes} x {a.m}, {a.epochs} epochs x {a.iterations} walk(s) a node: default (resident "
      f"cells) vs walk-ordered / global negatives: {res}", flush=True)

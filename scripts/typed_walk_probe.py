"""Walk-sampler throughput with and without typed transitions (BA 10 M / 100 M, 2^19 walks of
128 steps per launch).  Usage: python scripts/typed_walk_probe.py [--nodes N]"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import embiggen_amd as E  # noqa: E402
from embiggen_amd import ops  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--nodes", type=int, default=10_000_000)
ap.add_argument("--walks", type=int, default=1 << 19)
a = ap.parse_args()

g = E.barabasi_albert(a.nodes, 10, seed=42)
dev = torch.device("cuda", 0)
gen = torch.Generator(device=dev).manual_seed(1)
nt = torch.randint(0, 4, (g.get_number_of_nodes(),), generator=gen, device=dev, dtype=torch.int32)
et = torch.randint(0, 8, (g.get_number_of_directed_edges(),), generator=gen, device=dev,
                   dtype=torch.int32)
typed = g.with_types(nt, et)


def rate(graph, wp, label):
    ops.walks(graph, wp, 1, 0, 0, a.walks)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(3):
        ops.walks(graph, wp, 1, 1 + i, 0, a.walks)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 3
    print(f"{label:46s} {a.walks * 127 / dt:.3e} steps/s  ({dt * 1e3:.1f} ms)", flush=True)


rate(g, ops.walk_params(128, 10, 0.25, 4.0), "untyped rw.25/ew4")
rate(typed, ops.walk_params(128, 10, 0.25, 4.0), "typed graph, unit type weights rw.25/ew4")
rate(g, ops.walk_params(128, 10, 1.0, 1.0), "untyped first order")
rate(typed, ops.walk_params(128, 10, 1.0, 1.0, 100, 2.0, 1.0), "first order + node type x2")
rate(typed, ops.walk_params(128, 10, 1.0, 1.0, 100, 1.0, 0.5), "first order + edge type x0.5")
rate(typed, ops.walk_params(128, 10, 0.25, 4.0, 100, 2.0, 0.5), "rw.25/ew4 + node x2 + edge x0.5")
rate(g, ops.walk_params(128, 10, 2.0, 0.5), "untyped rw2/ew.5")
rate(typed, ops.walk_params(128, 10, 2.0, 0.5, 100, 2.0, 0.5), "rw2/ew.5 + node x2 + edge x0.5")
rate(g, ops.walk_params(128, 10, 4.0, 0.25), "untyped rw4/ew.25")
rate(g, ops.walk_params(128, 10, 0.5, 2.0), "untyped rw.5/ew2")
# round 6: max_neighbours (node2vec_skipgram.py:22,78-81) -- every line above ran at the
# reference's default of 100 (steps out of hubs choose among a per-visit sub-sample of 100 edges);
# the exact walks (None) and the smoke configuration's 10 beside it
for mn in (None, 10):
    rate(g, ops.walk_params(128, 10, 0.25, 4.0, mn), f"untyped rw.25/ew4, max_neighbours {mn}")
    rate(g, ops.walk_params(128, 10, 1.0, 1.0, mn), f"untyped first order, max_neighbours {mn}")
    rate(g, ops.walk_params(128, 10, 2.0, 0.5, mn), f"untyped rw2/ew.5, max_neighbours {mn}")
    rate(typed, ops.walk_params(128, 10, 0.25, 4.0, mn, 2.0, 0.5),
         f"rw.25/ew4 + node x2 + edge x0.5, max_neighbours {mn}")

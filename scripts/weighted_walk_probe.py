"""Walk-sampler throughput on a weighted graph (BA 10 M / 100 M, random weights)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import embiggen_amd as E
from embiggen_amd import ops
g = E.barabasi_albert(10_000_000, 10, seed=42)
wg = g.with_degree_normalized_weights()  # normalize_by_degree=True: a weighted graph
def rate(graph, wp, label, n=1 << 19):
    ops.walks(graph, wp, 1, 0, 0, n); torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(3): ops.walks(graph, wp, 1, 1 + i, 0, n)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 3
    print(f"{label:40s} {n * 127 / dt:.3e} steps/s ({dt * 1e3:.1f} ms)", flush=True)
if wg is not None:
    rate(wg, ops.walk_params(128, 10, 1.0, 1.0), "weighted first order")
    rate(wg, ops.walk_params(128, 10, 0.25, 4.0), "weighted rw.25/ew4")
    rate(wg, ops.walk_params(128, 10, 2.0, 0.5), "weighted rw2/ew.5")
rate(g, ops.walk_params(128, 10, 0.25, 4.0), "unweighted rw.25/ew4")

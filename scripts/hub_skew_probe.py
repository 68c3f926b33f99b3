"""Largest in-degree of the test graphs against their edges (the hub-skew rule of
gn2v_block_auto_plan_graph) and the plans that follow."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import embiggen_amd as E
from embiggen_amd.distributed import GpuBlockBackend
for n, m, seed in ((100_000, 5, 42), (120_000, 5, 3), (169_343, 7, 42), (200_000, 8, 42), (400_000, 5, 42), (1_000_000, 10, 42), (2_449_029, 25, 42), (10_000_000, 10, 42)):
    g = E.barabasi_albert(n, m, seed)
    t = g._device_tensors
    deg = (t["row_ptr"][1:] - t["row_ptr"][:-1])
    e = int(t["row_ptr"][-1])
    hub = int(deg.max())
    b = GpuBlockBackend(g, "cuda:0")
    print(n, m, "edges", e, "max degree", hub, "hub x 256 / edges", round(hub * 256 / e, 3), "plan d=128", b.auto_plan(1, 128, 10), "d=32", b.auto_plan(1, 32, 10), flush=True)
    del g, t, deg, b
    torch.cuda.empty_cache()

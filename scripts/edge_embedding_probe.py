"""gn2v_edge_embedding (the operators of embedding_transformers/edge_transformer.py on
the device, fused with the row gather) measured against its HBM roofline: BA graph, d = 128, the
graph's own directed edges as the edge list; algorithmic bytes per edge = two rows of 4 d B read +
the operator's output written (+ 8 B of ids)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

import embiggen_amd as E
from embiggen_amd import ops
from embiggen_amd.embedding_transformers import edge_embedding

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
g = E.barabasi_albert(n, 10, 42)
d = 128
table = ops.init_table(n, d, 42, 0, d ** -0.5)
row_ptr = torch.from_numpy(np.asarray(g.row_ptr).astype(np.int64)).cuda()
dst = torch.from_numpy(np.asarray(g.col_idx).astype(np.int32)).cuda()
src = torch.repeat_interleave(torch.arange(n, device="cuda", dtype=torch.int32),
                              (row_ptr[1:] - row_ptr[:-1]))
m = min(src.numel(), 1 << 27)  # 134 M edges: Concatenate writes 137 GB at d = 128
src, dst = src[:m].contiguous(), dst[:m].contiguous()
perm = torch.randperm(m, device="cuda")
rsrc, rdst = src[perm].contiguous(), dst[perm].contiguous()
print(f"BA {n} nodes, {m} directed edges, d = {d}, table {table.numel() * 4 / 1e9:.2f} GB")
for method in ("Hadamard", "Concatenate", "L2", "CosineSimilarity", "L2Distance"):
    for name, (s_, d_) in (("edges in CSR order", (src, dst)), ("edges shuffled", (rsrc, rdst))):
        width = 2 * d if method == "Concatenate" else 1 if method in ("CosineSimilarity", "L2Distance") else d
        if m * width * 4 > 150e9:
            s_, d_ = s_[: m // 2], d_[: m // 2]
        del_me = edge_embedding(table, s_, d_, method)  # the allocator's first block of this size
        del del_me
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        out = edge_embedding(table, s_, d_, method)
        b.record()
        torch.cuda.synchronize()
        ms = a.elapsed_time(b)
        e = s_.numel()
        algo = e * (2 * d * 4 + width * 4 + 8)
        print(f"{method:17s} {name:19s}: {e / ms * 1e3:.3e} edges/s, {algo / ms / 1e6:.0f} GB/s algorithmic "
              f"= {algo / ms / 1e6 / 8000:.2f} of 8 TB/s ({ms:.1f} ms)", flush=True)
        del out

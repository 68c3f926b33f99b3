"""CBOW, atomics vs write-through stores on BA graphs of 8 k - 64 k nodes at d = 32 / 64 / 128:
where do stores reach the link AUROC of atomics?  (The graph family with hubs is the hard one.)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch

import embiggen_amd as E
from sharded_helpers import link_auc_device

gen = torch.Generator(device="cuda")
for n in (8192, 16384, 32768, 65536):
    g = E.barabasi_albert(n, 5, 42)
    for d in (32, 64, 128):
        row = []
        for mode in ("atomic", "write_through"):
            m = E.models.CBOW(embedding_size=d, epochs=5, update_mode=mode, verbose=False)
            c, x, st = m.fit_transform_device(g)
            gen.manual_seed(1)
            row.append((link_auc_device(g, x[:, :d], c[:, :d], gen), st["train_ms"]))
        print(f"BA {n:6d} d={d:4d} atomic {row[0][0]:.4f} ({row[0][1]:6.0f} ms)  stores {row[1][0]:.4f} "
              f"({row[1][1]:6.0f} ms)  diff {row[1][0] - row[0][0]:+.4f}  speed x{row[0][1] / row[1][1]:.2f}",
              flush=True)

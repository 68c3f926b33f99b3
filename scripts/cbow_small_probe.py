"""CBOW on small graphs: update modes of the walk-ordered kernel (atomics on every row are the
default below 2^16 nodes) -- throughput and link AUROC of whole fits on the same walks.
    python scripts/cbow_small_probe.py NODES M EPOCHS"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "scripts"))
import torch

import embiggen_amd as E
from quality_probe import evaluate

nodes, m, epochs = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
g = E.barabasi_albert(nodes, m, 42)
gen = torch.Generator(device="cuda")
for mode in ("atomic", "write_through", "write_back"):
    model = E.models.CBOW(embedding_size=128, epochs=epochs, update_mode=mode, verbose=False)
    t0 = time.time()
    c, x, st = model.fit_transform_device(g)
    torch.cuda.synchronize()
    dt = time.time() - t0
    gen.manual_seed(1)
    # CBOW: the contextual table is the input side (what the wrapper returns first)
    auc_cx, auc_cos = evaluate(g, x[:, :128], c[:, :128], 200000, gen)
    print(f"BA {nodes} x {m} {epochs} epochs CBOW {mode:14s} {dt:6.2f}s train_ms {st['train_ms']:.0f} "
          f"({st['pairs'] / st['train_ms'] * 1e3:.3e} pairs/s) AUC(c.x)={auc_cx:.4f} "
          f"AUC(cos input side)={auc_cos:.4f} finite="
          f"{bool(torch.isfinite(c).all() and torch.isfinite(x).all())} |c|max={float(c.abs().max()):.2f} "
          f"|x|max={float(x.abs().max()):.2f}", flush=True)

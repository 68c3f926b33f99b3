"""Where do the racy store modes start to cost quality?  Link AUROC of atomic vs write-through on
BA graphs of growing size (same walks, 10 per node), SkipGram and CBOW."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import embiggen_amd as E
from embiggen_amd import ops, _lib
from sharded_helpers import link_auc_device as auc
d = 64
gen = torch.Generator(device="cuda")
for n in (1024, 4096, 16384, 65536, 262144):
    g = E.barabasi_albert(n, 8, 42)
    wp = ops.walk_params(64, 10, 1.0, 1.0)
    total = n * 10
    for model, name in ((0, "sgns"), (1, "cbow")):
        out = []
        for fl, label in ((_lib.TRAIN_ATOMIC, "atomic"), (_lib.TRAIN_WRITE_THROUGH, "wt")):
            tp = ops.train_params(model, d, 5, 4, flags=1 | fl)
            c = ops.init_table(n, d, 42, 0, d ** -0.5); x = ops.init_table(n, d, 42, 1, d ** -0.5)
            step = ops.sgns_step if model == 0 else ops.cbow_step
            for ep in range(3):
                for first in range(0, total, 1 << 16):
                    nb = min(1 << 16, total - first)
                    step(g, tp, ops.walks(g, wp, 42, ep, first, nb), 42, ep, first, 0.025, c, x)
            gen.manual_seed(1)
            a = auc(g, c, x, gen, 50000) if model == 0 else auc(g, x, c, gen, 50000)
            out.append(f"{label} {a:.4f}")
        print(f"n={n:7d} {name}: " + "  ".join(out), flush=True)

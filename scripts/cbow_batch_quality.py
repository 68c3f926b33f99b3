"""CBOW over several GPUs, measured before it is refused or built.

What any scheme that never shares a row must do for CBOW: a centre needs the mean of its <= 2w
CONTEXTUAL rows, which the block partition spreads over all parts.  The three-phase scheme of
DESIGN.md 8 (window sums while the contextual parts rotate, output rows trained against the frozen
means, gradients scattered in a second rotation) and the cheaper variant (contextual rows fetched
read-only, their deltas returned once per part episode) have the same numerical content: during a
"batch" of walks the contextual table is READ-ONLY and the input-side gradients of all the batch's
centres are summed and applied at its end; the central (output) table is trained in place as
always (its rows never leave their rank).  That content is what this script runs on one GPU
through the product kernel (gn2v_step_io.d_context_delta): link AUROC against the batch length,
next to the ordinary one-GPU CBOW on the same walks.  The batch length a fabric could sustain is
printed beside it: two rotations of the contextual table (n x 512 B per rank) must hide behind
the batch's training (3.3e8 centres/s per GPU)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch

import embiggen_amd as E
from embiggen_amd import ops
from sharded_helpers import link_auc_device as _auc

nodes = int(sys.argv[1]) if len(sys.argv) > 1 else 200_000
total = int(sys.argv[2]) if len(sys.argv) > 2 else 1 << 21
d, L, w, k, lr = 64, 64, 4, 5, 0.025
g = E.barabasi_albert(nodes, 8, 42)
n = g.get_number_of_nodes()
wp = ops.walk_params(L, 1, 1.0, 1.0)
tp = ops.train_params(1, d, k, w, flags=1)
gen = torch.Generator(device="cuda")
launch = 1 << 16


def fresh():
    return ops.init_table(n, d, 42, 0, d ** -0.5), ops.init_table(n, d, 42, 1, d ** -0.5)


def score(c, x):
    gen.manual_seed(1)
    # CBOW: input side = contextual, output side = central; same symmetrised score as SkipGram
    return round(_auc(g, c, x, gen), 4)


c, x = fresh()
for first in range(0, total, launch):
    nb = min(launch, total - first)
    ops.cbow_step(g, tp, ops.walks(g, wp, 42, 0, first, nb), 42, 0, first, lr, c, x)
print(f"BA {n} nodes, d {d}, {total} walks of {L}: one trainer (in place)", score(c, x), flush=True)

# walks per batch and rank a fabric hides: 2 rotations of n * 4d bytes at ~60 GB/s per link
# direction behind batch * L centres at 3.3e8 centres/s
sustain = int(2 * n * 4 * 128 / 60e9 * 3.3e8 / 128)
print(f"batch a rank of 8 could hide two rotations behind (d = 128): >= {sustain} walks per rank "
      f"= {8 * sustain} walks per job batch", flush=True)
import os
if os.environ.get("GN2V_CBOW_LAZY") is not None:
    raise SystemExit(0)
for batch in (1 << 8, 1 << 10, 1 << 12, 1 << 14, 1 << 16, 1 << 18):
    for scale in (1.0, "mean"):
        c, x = fresh()
        delta = torch.zeros_like(x)
        touched = torch.zeros(n, 1, device="cuda")
        for b0 in range(0, total, batch):
            nb_total = min(batch, total - b0)
            delta.zero_()
            for first in range(b0, b0 + nb_total, launch):
                nb = min(launch, b0 + nb_total - first)
                wk = ops.walks(g, wp, 42, 0, first, nb)
                ops.step(g, tp, wk, 42, 0, first, lr, c, x, context_delta=delta)
                if scale == "mean":
                    touched.zero_()
            if scale == "mean":
                # the sum of hundreds of gradients taken at one stale value overshoots: divide by
                # the number of centres that contributed (a per-row mean step)
                wk_all = ops.walks(g, wp, 42, 0, b0, nb_total).long().flatten()
                cnt = torch.bincount(wk_all, minlength=n).float().clamp_min(1.0).unsqueeze(1)
                x += delta / cnt.sqrt()
            else:
                x += delta
        label = "sum" if scale == 1.0 else "sum / sqrt(visits)"
        print(f"batch {batch:7d} walks, contextual += {label}", score(c, x),
              "finite" if bool(torch.isfinite(x).all()) else "NOT FINITE", flush=True)

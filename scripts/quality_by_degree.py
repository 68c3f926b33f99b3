"""Where does the default schedule's distance to the reference-semantic fit sit: on pairs that
touch high-degree nodes (cell-local negatives under-draw a hub: its cell's total degree is mostly
its own) or everywhere (the limited set of cell-mates)?  Config 3's shape, the quality gates' fits;
mean |d cos| and Spearman per bucket of the pair's larger degree, next to the floor (the
reference-semantic fit against itself under other negatives)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch

import embiggen_amd as E
from embiggen_amd import _lib, models
import test_gpu_quality_gates as Q

nodes, m, rw, ew, iterations, epochs, ref_mode = Q.SHAPES["config3_arxiv_shape"]
g = E.barabasi_albert(nodes, m, 42)
kw = dict(embedding_size=128, epochs=epochs, walk_length=128, iterations=iterations,
          window_size=5, number_of_negative_samples=10, return_weight=rw, explore_weight=ew,
          learning_rate=0.01, random_state=42, verbose=False)
n_each = 200_000
u, v = Q._pairs(g, n_each, 7)
deg = torch.from_numpy(g.get_node_degrees().astype(np.int64)).cuda()
big = torch.maximum(deg[u], deg[v])

fast = models.SkipGram(**kw)
c_fast, x_fast, _ = fast.fit_transform_device(g)
cos_fast = Q._cos(c_fast, u, v)
norm_fast = c_fast.norm(dim=1)
del c_fast, x_fast
ref = models.SkipGram(block_path=False, update_mode=ref_mode, **kw)
c_ref, x_ref, _ = ref.fit_transform_device(g)
cos_ref = Q._cos(c_ref, u, v)
norm_ref = c_ref.norm(dim=1)
del c_ref, x_ref
c_other = Q._reference_fit_with_other_negatives(g, kw, _lib.TRAIN_ATOMIC, 43)
cos_other = Q._cos(c_other, u, v)
norm_other = c_other.norm(dim=1)
del c_other
print(f"all pairs: spearman default/ref {Q._spearman(cos_fast, cos_ref):.4f} floor {Q._spearman(cos_ref, cos_other):.4f}; "
      f"mean|dcos| {float((cos_fast - cos_ref).abs().mean()):.4f} floor {float((cos_ref - cos_other).abs().mean()):.4f}")
for lo, hi in ((0, 20), (20, 50), (50, 200), (200, 1000), (1000, 10 ** 9)):
    for name, sl in (("edges", slice(0, n_each)), ("random", slice(n_each, 2 * n_each))):
        mk = (big[sl] >= lo) & (big[sl] < hi)
        if int(mk.sum()) < 50:
            continue
        a, b, c = cos_fast[sl][mk], cos_ref[sl][mk], cos_other[sl][mk]
        print(f"max degree [{lo:5d}, {hi:10d}) {name:6s} n {int(mk.sum()):7d}: mean cos default {float(a.mean()):+.4f} "
              f"ref {float(b.mean()):+.4f} other {float(c.mean()):+.4f}; mean|dcos| {float((a - b).abs().mean()):.4f} "
              f"floor {float((b - c).abs().mean()):.4f}; signed {float((a - b).mean()):+.4f} floor {float((c - b).mean()):+.4f}")
for lo, hi in ((0, 20), (20, 50), (50, 200), (200, 1000), (1000, 10 ** 9)):
    mk = (deg >= lo) & (deg < hi)
    print(f"degree [{lo:5d}, {hi:10d}) rows {int(mk.sum()):7d}: |central row| default {float(norm_fast[mk].mean()):.4f} "
          f"ref {float(norm_ref[mk].mean()):.4f} other {float(norm_other[mk].mean()):.4f}")

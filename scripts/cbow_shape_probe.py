"""CBOW away from the bench point: whole fits (one epoch over BA 1 M, one walk per node) at the
given "d:window" shapes through the model class; prints centres/s and the kernel's fraction of the
HBM roofline by algorithmic bytes (2 (c + 1 + k) rows per centre).  The kernel that
runs follows the dispatch rule of gn2v_api.hip (lazy window while a CU's LDS holds
kLazyMinWaves of its waves, else the uncached kernel); GN2V_CBOW_LAZY_WAVES /
GN2V_CBOW_LAZY_MIN_WAVES / GN2V_CBOW_LAZY=0 in the environment force the alternatives.

    python scripts/cbow_shape_probe.py 128:5 128:12 512:5
"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import embiggen_amd as E  # noqa: E402

shapes = [tuple(int(v) for v in s.split(":")) for s in sys.argv[1:]] or [(128, 5)]
g = E.barabasi_albert(1_000_000, 10, 42)
tag = " ".join(f"{k}={v}" for k, v in os.environ.items() if k.startswith("GN2V_CBOW"))
for d, w in shapes:
    kw = dict(embedding_size=d, epochs=1, iterations=1, walk_length=128, window_size=w,
              number_of_negative_samples=10, return_weight=0.25, explore_weight=4.0, verbose=False)
    m = E.models.CBOW(**kw)
    m.fit_transform_device(g, max_walks_per_epoch=1 << 16)  # warm-up
    torch.cuda.synchronize()
    t0 = time.time()
    c, x, st = m.fit_transform_device(g)
    torch.cuda.synchronize()
    dt = time.time() - t0
    row = m.padded_size * 4
    centres = 1_000_000 * 128
    byts = st["pairs"] * 2 * row + centres * 2 * 11 * row
    print(f"cbow d={d:4d} w={w:2d} [{tag}] {centres / dt:.3e} centres/s, train {st['train_ms']:7.1f} ms, "
          f"frac(kernel) {byts / (st['train_ms'] * 1e-3) / 8e12:.3f}, finite "
          f"{bool(torch.isfinite(c).all() and torch.isfinite(x).all())}", flush=True)

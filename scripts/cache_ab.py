"""Same-process A/B of the SkipGram kernel with and without the LDS context cache."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
import embiggen_amd as E
from embiggen_amd import _lib, ops
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
g = E.barabasi_albert(n, 10, 42); d = 128
wk = ops.walks(g, ops.walk_params(128, 10, 0.25, 4.0), 42, 0, 0, 1 << 16)
variants = {"plain": _lib.TRAIN_WRITE_THROUGH | _lib.TRAIN_NO_CTX_CACHE,
            "cache(degree rule)": _lib.TRAIN_WRITE_THROUGH,
            "cache(all rows)": _lib.TRAIN_WRITE_THROUGH | _lib.TRAIN_CTX_CACHE_ALL}
c = ops.init_table(n, d, 42, 0, d ** -0.5); x = ops.init_table(n, d, 42, 1, d ** -0.5)
for rep in range(3):
    for name, fl in variants.items():
        tp = ops.train_params(0, d, 10, 5, flags=1 | fl)
        ops.stats_reset(g); ops.sgns_step(g, tp, wk, 42, 0, 0, 0.01, c, x); st = ops.stats_read(g)
        print(f"rep{rep} {name:20s} {st['train_ms']:.1f} ms  {st['pairs']/st['train_ms']*1e3:.3e} pairs/s  frac {st['pairs']*12288/(st['train_ms']*1e-3)/8e12:.3f}", flush=True)

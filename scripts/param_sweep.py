"""Throughput of whole fits over the parameters bench.py keeps fixed (negatives, window, walk
length, p / q), through the model classes on BA 1 M: looks for cliffs away from the defaults.
Fraction = algorithmic bytes (SkipGram 2 (k + 2) rows per pair, CBOW 2 (c + 1 + k)
rows per centre) / time / 8 TB/s."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

import embiggen_amd as E

g = E.barabasi_albert(1_000_000, 10, 42)
base = dict(embedding_size=128, epochs=1, iterations=1, walk_length=128, window_size=5,
            number_of_negative_samples=10, return_weight=0.25, explore_weight=4.0, verbose=False)
variants = [{}, {"number_of_negative_samples": 1}, {"number_of_negative_samples": 5},
            {"number_of_negative_samples": 20}, {"number_of_negative_samples": 50},
            {"window_size": 1}, {"window_size": 2}, {"window_size": 10}, {"window_size": 20},
            {"walk_length": 16}, {"walk_length": 64}, {"walk_length": 512},
            {"return_weight": 1.0, "explore_weight": 1.0}, {"return_weight": 4.0, "explore_weight": 0.25},
            {"use_scale_free_distribution": False}, {"stochastic_downsample_by_degree": True},
            {"normalize_learning_rate_by_degree": True}]
for cls in (E.models.SkipGram, E.models.CBOW):
    for v in variants:
        kw = dict(base, **v)
        m = cls(**kw)
        m.fit_transform_device(g, max_walks_per_epoch=1 << 16)  # warm-up (allocations, alias tables)
        torch.cuda.synchronize()
        t0 = time.time()
        c, x, st = m.fit_transform_device(g)
        torch.cuda.synchronize()
        dt = time.time() - t0
        k, w, L = kw["number_of_negative_samples"], kw["window_size"], kw["walk_length"]
        row = 128 * 4
        if cls is E.models.SkipGram:
            units, byts = st["pairs"], st["pairs"] * 2 * (k + 2) * row
        else:
            units = 1_000_000 * L
            byts = st["pairs"] * 2 * row + units * 2 * (k + 1) * row  # contexts + output rows
        print(f"{cls.__name__:8s} {str(v):58s} {dt:6.2f}s train {st['train_ms']:7.0f} ms walk "
              f"{st['walk_ms']:5.0f} ms  {units / dt:.3e} {'pairs' if cls is E.models.SkipGram else 'centres'}/s "
              f"frac(kernel) {byts / (st['train_ms'] * 1e-3) / 8e12:.3f} plan {m.last_plan is not None} "
              f"finite {bool(torch.isfinite(c).all() and torch.isfinite(x).all())}", flush=True)

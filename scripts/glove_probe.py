"""GloVe path on a scale-free graph: co-occurrence build time and the SGD kernel's rate against
its HBM roofline (algorithmic bytes per entry = (2 + 2 / 16) * d * 4: the contextual row read and
written per entry, the central row once per record of 16 entries).
Usage: python scripts/glove_probe.py [--nodes N] [--walk-length L] [--d D]"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import embiggen_amd as E  # noqa: E402
from embiggen_amd import _lib, cooccurrence, models, ops  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--nodes", type=int, default=1_000_000)
ap.add_argument("--walk-length", type=int, default=128)
ap.add_argument("--window", type=int, default=5)
ap.add_argument("--d", type=int, default=128)
ap.add_argument("--epochs", type=int, default=5)
a = ap.parse_args()

g = E.barabasi_albert(a.nodes, 10, seed=42)
m = models.GloVe(embedding_size=a.d, walk_length=a.walk_length, window_size=a.window,
                 iterations=1, verbose=False)
torch.cuda.synchronize()
t0 = time.perf_counter()
keys, counts = m.cooccurrence_device(g)
torch.cuda.synchronize()
t1 = time.perf_counter()
rows, cols, logx, fx = cooccurrence.entries(keys, counts, 42, 0.75)
torch.cuda.synchronize()
t2 = time.perf_counter()
slots = a.nodes * a.walk_length * 2 * a.window
n_entries = int((cols != -1).sum())
print(f"co-occurrence: {slots:.3e} slots -> {n_entries:.3e} entries in {rows.numel() // 16} records, {t1 - t0:.2f}s "
      f"(+ {t2 - t1:.2f}s ordering)", flush=True)
del keys, counts
n, d, ld = a.nodes, a.d, m.padded_size


def loss_of(central, contextual, bc, bx):
    """GloVe loss of the entries, in slices (torch on the device; measurement only)."""
    total = 0.0
    step = 1 << 24
    for lo in range(0, rows.numel(), step):
        r, c = rows[lo:lo + step].long(), cols[lo:lo + step].long()
        ok = c >= 0
        r, c = r[ok], c[ok]
        diff = (central[r] * contextual[c]).sum(1) + bc[r] + bx[c] - logx[lo:lo + step][ok]
        total += float((0.5 * fx[lo:lo + step][ok] * diff * diff).sum())
    return total


for label, flags in (("write-through", _lib.TRAIN_WRITE_THROUGH), ("write-back", _lib.TRAIN_WRITE_BACK),
                     ("atomic", _lib.TRAIN_ATOMIC)):
    central = ops.init_table(n, d, 42, 0, d ** -0.5, 0, ld)
    contextual = ops.init_table(n, d, 42, 1, d ** -0.5, 0, ld)
    bc = torch.zeros(n, dtype=torch.float32, device="cuda")
    bx = torch.zeros(n, dtype=torch.float32, device="cuda")
    ops.glove_step(g, rows, cols, logx, fx, central, contextual, bc, bx, d, 0.05, flags)
    torch.cuda.synchronize()
    ops.stats_reset(g, 0)
    t0 = time.perf_counter()
    for _ in range(a.epochs):
        ops.glove_step(g, rows, cols, logx, fx, central, contextual, bc, bx, d, 0.05, flags)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / a.epochs
    st = ops.stats_read(g, 0)
    kernel = st["train_ms"] / max(st["train_launches"], 1) * 1e-3
    bytes_per = (2 + 2 / 16) * d * 4
    print(f"{label:14s} {n_entries / dt:.3e} entries/s wall, kernel {kernel * 1e3:.1f} ms -> "
          f"{n_entries * bytes_per / kernel / 1e9:.0f} GB/s algorithmic "
          f"({n_entries * bytes_per / kernel / 8e12:.2f} of 8 TB/s), finite "
          f"{bool(torch.isfinite(central).all())}, loss after {a.epochs + 1} epochs "
          f"{loss_of(central, contextual, bc, bx):.4e}", flush=True)

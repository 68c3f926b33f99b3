"""Does giving every XCD its own slice of the table rows pay?  Random-row read-modify-write with
row ids drawn like degree-proportional negatives of a BA graph (density ~ 1/sqrt(row)), on tables
of the sizes a context part has at N = 8 / 1 GPUs:
  A  ids in random order, write-through (sc1) stores   -- today's policy, any XCD touches any row
  B  ids arranged so that workgroup b only touches rows with row % 8 == b % 8, plain stores
     (block b runs on XCD b % 8: every row then lives in exactly one XCD's L2, so write-back
      caching is safe and hub rows stay L2 resident)
  C  the arrangement of B with write-through stores (separates locality from store policy)
  D  ids in random order, plain stores (racy across XCDs; speed reference only)
Uses the calibration kernel gn2v::touch_rows_kernel (row i is handled by block (i / 16) % grid)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

from embiggen_amd import _lib, ops

d = 128
n_ids = 1 << 25


def run(label, fn, bytes_, reps=5):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    print(f"{label:58s} {dt * 1e3:8.3f} ms  {bytes_ / dt / 1e12:6.3f} TB/s", flush=True)


def sliced(ids, slices=8):
    """Reorder ids so that chunk c of 16 consecutive ids holds rows with row % slices == c % slices."""
    per = []
    m = None
    for s in range(slices):
        sel = ids[ids % slices == s]
        per.append(sel)
        m = sel.numel() if m is None else min(m, sel.numel())
    m = m // 16 * 16
    stacked = torch.stack([p[:m].reshape(-1, 16) for p in per], dim=1)  # [chunks, slices, 16]
    return stacked.reshape(-1).contiguous()


for rows, dist in ((625_000, "ba"), (1_250_000, "ba"), (10_000_000, "ba"), (10_000_000, "uniform"),
                   (625_000, "uniform")):
    t = torch.zeros((rows, d), dtype=torch.float32, device="cuda")
    g = torch.Generator(device="cuda")
    g.manual_seed(1)
    u = torch.rand(n_ids, device="cuda", generator=g, dtype=torch.float64)
    ids = ((u * u if dist == "ba" else u) * rows).long().clamp(max=rows - 1)
    # hubs have low ids in BA graphs and rows are striped over parts, so row order = degree order
    rnd = ids.to(torch.int32).contiguous()
    arr = sliced(ids).to(torch.int32).contiguous()
    blocks = torch.arange(arr.numel(), device="cuda") // 16 % 8
    assert bool((arr.long() % 8 == blocks).all())
    print(f"--- table {rows} rows x {d} f32 = {rows * d * 4 / 1e6:.0f} MB, ids {dist}", flush=True)
    for label, x, flag in (("A random order, write-through", rnd, _lib.TRAIN_WRITE_THROUGH),
                           ("B XCD-sliced order, write-back (plain stores)", arr, _lib.TRAIN_WRITE_BACK),
                           ("C XCD-sliced order, write-through", arr, _lib.TRAIN_WRITE_THROUGH),
                           ("D random order, write-back", rnd, _lib.TRAIN_WRITE_BACK)):
        run(label, lambda x=x, flag=flag: ops.touch_rows(t, x, flag), 2 * x.numel() * d * 4)
    del t, ids, rnd, arr, u

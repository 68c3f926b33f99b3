"""Pair-list kernel vs centre-record kernel ([centre, up to C contexts] per wave) on one GPU."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
import embiggen_amd as E
from embiggen_amd import _lib, ops
g = E.barabasi_albert(10_000_000, 10, 42); n, d = 10_000_000, 128
wp = ops.walk_params(128, 10, 0.25, 4.0)
wk = ops.walks(g, wp, 42, 0, 0, 1 << 17)
c = ops.init_table(n, d, 42, 0, d ** -0.5); x = ops.init_table(n, d, 42, 1, d ** -0.5)
pairs = ops.walk_pairs(wk, 5, 1)
pairs = pairs[torch.randperm(pairs.shape[0], device="cuda")].contiguous()
L, w, C = 128, 5, 10
idx = torch.arange(L, device="cuda")
offs = torch.tensor([-5, -4, -3, -2, -1, 1, 2, 3, 4, 5], device="cuda")
pos = idx[:, None] + offs[None, :]                                   # [L, 10]
ok = (pos >= 0) & (pos < L)
ctx = torch.where(ok[None], wk[:, pos.clamp(0, L - 1)], torch.full((1,), -1, dtype=torch.int32, device="cuda"))  # [nw, L, 10]
# left-pack the valid contexts (sentinels must be a suffix)
order = torch.argsort((~ok).to(torch.int8), dim=1, stable=True)      # [L, 10]
ctx = torch.gather(ctx, 2, order[None].expand(ctx.shape[0], -1, -1))
rec = torch.cat([wk[:, :, None], ctx], dim=2).reshape(-1, 1 + C)
rec = rec[torch.randperm(rec.shape[0], device="cuda")].contiguous()
npairs = int((rec[:, 1:] != -1).sum())
print("pairs", pairs.shape[0], "records", rec.shape[0], "pairs in records", npairs)
def run(label, walks, tp, count):
    ops.step(g, tp, walks, 42, 0, 0, 0.01, c, x, pair_mode=True); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(3): ops.step(g, tp, walks, 42, 0, 0, 0.01, c, x, pair_mode=True)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 3
    print(f"{label:18s} {count / dt:.3e} pairs/s  {dt * 1e3:.1f} ms  ({count * 12288 / dt / 8e12:.3f} of roofline)", flush=True)
run("pair list", pairs, ops.train_params(0, d, 10, 1, flags=1), pairs.shape[0])
run("centre records", rec, ops.train_params(0, d, 10, C, flags=1), npairs)

"""Time the host-side stages of one block-trainer round on one GPU (world = 1)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
import embiggen_amd as E
from embiggen_amd import _lib, ops
g = E.barabasi_albert(10_000_000, 10, 42); d = 128
nw = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 18
wp = ops.walk_params(128, 10, 0.25, 4.0)
def T(label, fn, reps=3):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): out = fn()
    torch.cuda.synchronize(); print(f"{label:28s} {(time.perf_counter()-t0)/reps*1e3:8.1f} ms", flush=True); return out
wk = T("walks", lambda: ops.walks(g, wp, 42, 0, 0, nw))
pairs = T("walk_pairs (+compaction)", lambda: ops.walk_pairs(wk, 5, 1))
n = pairs.shape[0]; print("pairs", n)
world = 8
def keys():
    centre = pairs[:, 0].to(torch.int64) & 0xFFFFFFFF
    ctx = pairs[:, 1].to(torch.int64) & 0xFFFFFFFF
    block = (centre % world) * world + ctx % world
    idx = torch.arange(n, dtype=torch.int64, device=pairs.device)
    salt = (idx * 0x3C6EF35F + 12345) & 0x7FFFFFFF
    salt = ((salt ^ (salt >> 15)) * 0x2C1B3C6D) & 0x7FFFFFFF
    salt = ((salt ^ (salt >> 12)) * 0x297A2D39) & 0x7FFFFFFF
    salt = salt ^ (salt >> 15)
    return block * (1 << 31) + salt
key = T("keys", keys)
order = T("argsort", lambda: torch.argsort(key))
sp = T("gather pairs[order]", lambda: pairs[order])
T("rows = pairs // world", lambda: torch.div(sp.to(torch.int64) & 0xFFFFFFFF, world, rounding_mode="floor").to(torch.int32))
T("sort (values only)", lambda: torch.sort(key))
k32 = (key >> 31).to(torch.int32)
T("argsort int32 block only", lambda: torch.argsort(k32))

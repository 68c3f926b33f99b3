"""Random-row read-modify-write bandwidth of the calibration kernel: 1 vs 2 rounds in flight,
write-through vs write-back, and pure-read / streaming references via torch."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
from embiggen_amd import _lib, ops

n, d = 10_000_000, 128
t = torch.zeros((n, d), dtype=torch.float32, device="cuda")
perm = torch.randperm(n, device="cuda").to(torch.int32)
def run(label, fn, bytes_, reps=5):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    print(f"{label:40s} {dt*1e3:8.3f} ms  {bytes_/dt/1e12:6.3f} TB/s", flush=True)
rw = 2 * n * d * 4
run("touch write_through 1 round", lambda: ops.touch_rows(t, perm, _lib.TRAIN_WRITE_THROUGH), rw)
run("touch write_through 2 rounds", lambda: ops.touch_rows(t, perm, _lib.TRAIN_WRITE_THROUGH | 256), rw)
run("touch write_back 1 round", lambda: ops.touch_rows(t, perm, _lib.TRAIN_WRITE_BACK), rw)
run("touch write_back 2 rounds", lambda: ops.touch_rows(t, perm, _lib.TRAIN_WRITE_BACK | 256), rw)
run("touch atomic float4-shaped", lambda: ops.touch_rows(t, perm, _lib.TRAIN_ATOMIC), rw)
run("touch atomic lane-contiguous", lambda: ops.touch_rows(t, perm, _lib.TRAIN_ATOMIC | 512), rw)
small = perm[:200000].contiguous()
rs = 2 * 200000 * d * 4
run("small(200k rows) atomic float4-shaped", lambda: ops.touch_rows(t, small, _lib.TRAIN_ATOMIC), rs, 20)
run("small(200k rows) atomic contiguous", lambda: ops.touch_rows(t, small, _lib.TRAIN_ATOMIC | 512), rs, 20)
run("small(200k rows) write_through", lambda: ops.touch_rows(t, small, _lib.TRAIN_WRITE_THROUGH), rs, 20)
seq = torch.arange(n, device="cuda", dtype=torch.int32)
run("touch sequential rows write_through", lambda: ops.touch_rows(t, seq, _lib.TRAIN_WRITE_THROUGH), rw)
run("touch sequential rows 2 rounds", lambda: ops.touch_rows(t, seq, _lib.TRAIN_WRITE_THROUGH | 256), rw)
u = torch.empty_like(t)
run("torch copy_ (streaming r+w)", lambda: u.copy_(t), rw)
run("torch add_ in place", lambda: t.add_(1.0), rw)
idx = perm.long()
run("torch index_select rows (r + w)", lambda: torch.index_select(t, 0, idx, out=u), rw)

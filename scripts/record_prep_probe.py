"""Stage times of the centre-record preparation of one block-trainer round (one GPU, keys as for
world = 8)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
import embiggen_amd as E
from embiggen_amd import ops
g = E.barabasi_albert(10_000_000, 10, 42)
nw = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
world, C = 8, 10
wp = ops.walk_params(128, 10, 0.25, 4.0)
def T(label, fn):
    torch.cuda.synchronize(); t0 = time.perf_counter(); out = fn(); torch.cuda.synchronize()
    print(f"{label:34s} {(time.perf_counter() - t0) * 1e3:8.1f} ms", flush=True); return out
for rep in range(2):
    print("rep", rep)
    wk = T("walks", lambda: ops.walks(g, wp, 42, 0, 0, nw))
    slots, keys = T("pair kernel (keys)", lambda: ops.walk_pair_blocks(wk, 5, 1, world, 2 ** 64 - 1))
    keys, order = T("sort", lambda: torch.sort(keys, stable=True))
    n = int(torch.searchsorted(keys, torch.tensor([world * world << 32], device="cuda"))[0])
    grouped = T("gather slots[order[:n]]", lambda: slots[order[:n]])
    del slots, order
    k = keys[:n]
    def flags():
        s = torch.ones(n, dtype=torch.bool, device="cuda"); torch.ne(k[1:], k[:-1], out=s[1:]); return s
    start = T("run starts", flags)
    run_pos = T("nonzero", lambda: torch.nonzero(start).flatten())
    run_id = T("cumsum", lambda: torch.cumsum(start, 0) - 1)
    rank = T("rank", lambda: torch.arange(n, dtype=torch.int64, device="cuda") - run_pos[run_id])
    run_len = torch.diff(run_pos, append=torch.tensor([n], device="cuda"))
    recs = torch.div(run_len + (C - 1), C, rounding_mode="floor")
    first_rec = torch.cumsum(recs, 0) - recs
    rec_id = T("rec_id", lambda: first_rec[run_id] + torch.div(rank, C, rounding_mode="floor"))
    n_rec = int(recs.sum())
    records = T("alloc records", lambda: torch.full((n_rec, 1 + C), -1, dtype=torch.int32, device="cuda"))
    flat = records.view(-1)
    def scat():
        flat[rec_id * (1 + C)] = grouped[:, 0]
        flat[rec_id * (1 + C) + 1 + rank % C] = grouped[:, 1]
    T("two scatters", scat)
    rec_block = T("repeat_interleave", lambda: torch.repeat_interleave(k[run_pos] >> 32, recs))
    T("bincount", lambda: torch.bincount(rec_block, minlength=world * world))
    order2 = T("argsort records", lambda: torch.argsort((rec_block << 31) | (torch.arange(n_rec, device="cuda") * 2654435761 & 0x7FFFFFFF), stable=True))
    out = T("gather records", lambda: records[order2])
    print("pairs", n, "runs", run_pos.numel(), "records", n_rec, "fill", n / (n_rec * C))
    del wk, keys, grouped, k, start, run_pos, run_id, rank, run_len, recs, first_rec, rec_id, records, flat, rec_block, order2, out

import sys, time, torch
sys.path.insert(0, '/root/repo')
import embiggen_amd as E
from embiggen_amd import cooccurrence, models, ops
g = E.barabasi_albert(1_000_000, 10, 42)
wp = ops.walk_params(128, 1, 0.25, 4.0)
def T(label, fn):
    torch.cuda.synchronize(); t0 = time.perf_counter(); out = fn(); torch.cuda.synchronize()
    print(f"{label:34s} {(time.perf_counter()-t0)*1e3:8.1f} ms", flush=True); return out
nb = (1 << 27) // (128 * 10)
for rep in range(2):
    print("rep", rep)
    walks = T("walks", lambda: ops.walks(g, wp, 42, 0, 0, nb))
    keys, weights = T("cooc_slots", lambda: ops.cooc_slots(walks, 5, 1))
    sk, order = T("sort", lambda: torch.sort(keys))
    w2 = T("gather weights", lambda: weights[order])
    uniq, inv = T("unique_consecutive", lambda: torch.unique_consecutive(sk, return_inverse=True))
    sums = T("index_add", lambda: torch.zeros(uniq.numel(), dtype=torch.int64, device="cuda").index_add_(0, inv, w2))
    T("last", lambda: int(uniq[-1]))
    a = T("reduce_slots whole", lambda: cooccurrence.reduce_slots(keys, weights))
    b = T("merge a+a", lambda: cooccurrence.merge(a, a))
    print("entries", a[0].numel())
m = models.GloVe(embedding_size=128, walk_length=128, window_size=5, iterations=1, verbose=False)
for rep in range(2):
    out = T("cooccurrence_device (1 M nodes)", lambda: m.cooccurrence_device(g))
    print("entries", out[0].numel(), "allocated GB", torch.cuda.max_memory_allocated() / 1e9)

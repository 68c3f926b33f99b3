import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
import embiggen_amd as E
from sharded_helpers import link_auc_device
g = E.barabasi_albert(1_000_000, 10, 42)
for k in (10, 20, 30, 50, 100):
    m = E.models.SkipGram(embedding_size=128, epochs=1, iterations=1, walk_length=128, window_size=5,
                          number_of_negative_samples=k, return_weight=0.25, explore_weight=4.0, verbose=False)
    m.fit_transform_device(g, max_walks_per_epoch=1 << 16)
    torch.cuda.synchronize(); t0 = time.time()
    c, x, st = m.fit_transform_device(g)
    torch.cuda.synchronize(); dt = time.time() - t0
    gen = torch.Generator(device="cuda"); gen.manual_seed(1)
    print(f"k={k:3d} plan {m.last_plan['parts']}x{m.last_plan['slices']} {st['pairs'] / dt:.3e} pairs/s "
          f"({st['pairs'] * (k + 1) / dt:.3e} samples/s) link AUROC {link_auc_device(g, c, x, gen):.4f} "
          f"finite {bool(torch.isfinite(c).all() and torch.isfinite(x).all())}", flush=True)

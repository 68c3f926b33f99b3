"""Whole default-parameter fits through the public API (soak test: many launches, event folding,
DataFrame result, both models)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np, torch
import embiggen_amd as E
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000
only = set(sys.argv[2].split(",")) if len(sys.argv) > 2 else None  # class names to run
g = E.barabasi_albert(n, 7, 42)
for cls, kw in ((E.Node2VecSkipGramEnsmallen, {}), (E.Node2VecCBOWEnsmallen, {}),
                (E.WalkletsSkipGramEnsmallen, {"epochs": 3}), (E.Node2VecGloVeEnsmallen, {}),
                (E.DeepWalkGloVeEnsmallen, {"epochs": 20})):
    if only and cls.__name__ not in only:
        continue
    m = cls(**kw)
    t0 = time.time()
    res = m.fit_transform(g, return_dataframe=False)
    dt = time.time() - t0
    tabs = res.get_all_node_embedding()
    st = m.get_last_stats()
    print(cls.__name__, f"{dt:.1f}s", "pairs", st["pairs"], f"{st['pairs']/dt:.3e} pairs/s",
          "launches", st["train_launches"], "tables", len(tabs), tabs[0].shape,
          "finite", all(np.isfinite(t).all() for t in tabs),
          "mem GB", round(torch.cuda.max_memory_allocated() / 1e9, 2), flush=True)

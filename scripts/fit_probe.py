import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
import embiggen_amd as E
for n, epochs in ((1_000_000, 5), (10_000_000, 1)):
    g = E.barabasi_albert(n, 7 if n < 5e6 else 10, 42)
    m = E.Node2VecSkipGramEnsmallen(epochs=epochs, verbose=True)
    t0 = time.time()
    res = m.fit_transform(g, return_dataframe=False)
    dt = time.time() - t0
    st = m.get_last_stats()
    tabs = res.get_all_node_embedding()
    print(n, "epochs", epochs, f"{dt:.1f}s", f"{st['pairs']/dt:.3e} pairs/s incl. host copy", "kernel ms", round(st["train_ms"]),
          "plan", m._model.last_plan, "finite", all(np.isfinite(t).all() for t in tabs),
          "peak GB", round(torch.cuda.max_memory_allocated()/1e9, 1), flush=True)
    del g, m, res, tabs

"""Wall time of whole default-parameter fits on small graphs (launch-bound territory): Karate,
BA shaped like Cora (2 708 / 5 429), BA shaped like ogbn-arxiv (169 343 / 1 166 243)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np, torch
import embiggen_amd as E

graphs = [("karate", E.karate_club()), ("BA 2708 x 2", E.barabasi_albert(2708, 2, 42)),
          ("BA 169343 x 7", E.barabasi_albert(169343, 7, 42))]
for name, g in graphs:
    for cls in (E.Node2VecSkipGramEnsmallen, E.Node2VecCBOWEnsmallen, E.DeepWalkSkipGramEnsmallen):
        m = cls(embedding_size=128)
        t0 = time.time()
        res = m.fit_transform(g, return_dataframe=False)
        dt = time.time() - t0
        st = m.get_last_stats()
        tabs = res.get_all_node_embedding()
        print(f"{name:14s} {cls.__name__:28s} {dt:7.2f}s pairs {st['pairs']:.3e} "
              f"{st['pairs'] / dt:.3e} pairs/s launches {st['train_launches']} "
              f"train_ms {st['train_ms']:.0f} walk_ms {st['walk_ms']:.0f} "
              f"finite {all(np.isfinite(t).all() for t in tabs)}", flush=True)

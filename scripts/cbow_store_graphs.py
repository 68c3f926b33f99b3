"""CBOW below 2^16 nodes: atomics (today's default) against write-through stores on graph families
other than Barabasi-Albert -- small world, uniform random, ring of cliques, 2-d grid -- whole fits
on the same walks, link AUROC of the symmetrised score and of the cosine of the input-side table.
    python scripts/cbow_store_graphs.py"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

import embiggen_amd as E


def small_world(n, k, p, rng):
    src = np.repeat(np.arange(n), k // 2)
    dst = (src + np.tile(np.arange(1, k // 2 + 1), n)) % n
    rewire = rng.rand(len(src)) < p
    dst[rewire] = rng.randint(0, n, rewire.sum())
    return src, dst


def uniform_random(n, m, rng):
    return rng.randint(0, n, n * m), rng.randint(0, n, n * m)


def ring_of_cliques(n, size, rng):
    c = n // size
    i, j = np.triu_indices(size, 1)
    src = (np.arange(c)[:, None] * size + i[None, :]).ravel()
    dst = (np.arange(c)[:, None] * size + j[None, :]).ravel()
    bridge_s = np.arange(c) * size
    bridge_d = ((np.arange(c) + 1) % c) * size + 1
    return np.concatenate([src, bridge_s]), np.concatenate([dst, bridge_d])


def grid(n, rng):
    side = int(n ** 0.5)
    idx = np.arange(side * side).reshape(side, side)
    return (np.concatenate([idx[:, :-1].ravel(), idx[:-1, :].ravel()]),
            np.concatenate([idx[:, 1:].ravel(), idx[1:, :].ravel()]))


def auc(pos, neg):
    s = torch.cat([pos, neg])
    ranks = torch.empty_like(s)
    ranks[torch.argsort(s)] = torch.arange(1, s.numel() + 1, device=s.device, dtype=s.dtype)
    n1, n0 = pos.numel(), neg.numel()
    return float((ranks[:n1].sum() - n1 * (n1 + 1) / 2) / (n1 * n0))


def evaluate(g, a, b, rng, n_eval=200000):
    """a: input-side table, b: output-side table (CBOW: contextual, central)."""
    n = g.get_number_of_nodes()
    e = rng.randint(0, len(g.col_idx), n_eval)
    src = torch.from_numpy(np.searchsorted(g.row_ptr.astype(np.int64), e, side="right") - 1).cuda()
    dst = torch.from_numpy(g.col_idx[e].astype(np.int64)).cuda()
    ru = torch.from_numpy(rng.randint(0, n, n_eval)).cuda()
    rv = torch.from_numpy(rng.randint(0, n, n_eval)).cuda()
    score = lambda u, v: (a[u] * b[v]).sum(1) + (a[v] * b[u]).sum(1)  # noqa: E731
    cos = lambda u, v: (a[u] * a[v]).sum(1) / (a[u].norm(dim=1) * a[v].norm(dim=1)).clamp_min(1e-6)  # noqa: E731
    return auc(score(src, dst), score(ru, rv)), auc(cos(src, dst), cos(ru, rv))


rng = np.random.RandomState(0)
for n in (8192, 30000):
    families = {"small world k=10 p=0.1": small_world(n, 10, 0.1, rng),
                "uniform random m=5": uniform_random(n, 5, rng),
                "ring of cliques of 16": ring_of_cliques(n, 16, rng),
                "grid": grid(n, rng)}
    for name, (src, dst) in families.items():
        g = E.CSRGraph.from_edge_list(src, dst, number_of_nodes=n)
        for mode in ("atomic", "write_through"):
            m = E.models.CBOW(embedding_size=128, epochs=10, update_mode=mode, verbose=False)
            t0 = time.time()
            c, x, st = m.fit_transform_device(g)
            torch.cuda.synchronize()
            dt = time.time() - t0
            a_cx, a_cos = evaluate(g, x[:, :128], c[:, :128], np.random.RandomState(1))
            print(f"{n:6d} {name:24s} CBOW {mode:14s} {dt:6.2f}s train_ms {st['train_ms']:6.0f} "
                  f"AUC(c.x)={a_cx:.4f} AUC(cos)={a_cos:.4f} |c|max={float(c.abs().max()):.2f} "
                  f"|x|max={float(x.abs().max()):.2f}", flush=True)

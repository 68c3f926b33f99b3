"""Quality of the update modes at scale: train Node2Vec SkipGram on a BA graph with each mode on
identical walks and compare link-prediction AUROC (edges vs random pairs) computed on the GPU.
    python scripts/quality_probe.py --nodes 1000000 --walks 1000000
"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

import embiggen_amd as E  # noqa: E402
from embiggen_amd import _lib, ops  # noqa: E402


def auc(pos, neg):
    s = torch.cat([pos, neg])
    ranks = torch.empty_like(s)
    order = torch.argsort(s)
    ranks[order] = torch.arange(1, s.numel() + 1, device=s.device, dtype=s.dtype)
    n1, n0 = pos.numel(), neg.numel()
    return float((ranks[:n1].sum() - n1 * (n1 + 1) / 2) / (n1 * n0))


def evaluate(g, c, x, n_eval, gen):
    t = g._device_tensors
    if t is None:  # a graph built on the host
        import numpy as np

        t = {"row_ptr": torch.from_numpy(np.asarray(g.row_ptr).astype(np.int64)).cuda(),
             "col_idx": torch.from_numpy(np.asarray(g.col_idx).astype(np.int64)).cuda()}
    n = g.get_number_of_nodes()
    e = torch.randint(0, t["col_idx"].numel(), (n_eval,), device="cuda", generator=gen)
    dst = t["col_idx"][e].long()
    src = torch.searchsorted(t["row_ptr"], e, right=True) - 1
    ru = torch.randint(0, n, (n_eval,), device="cuda", generator=gen)
    rv = torch.randint(0, n, (n_eval,), device="cuda", generator=gen)

    def score(u, v):
        return (c[u] * x[v]).sum(1) + (c[v] * x[u]).sum(1)

    def cos(u, v):
        a, b = c[u], c[v]
        return (a * b).sum(1) / (a.norm(dim=1) * b.norm(dim=1)).clamp_min(1e-6)

    return auc(score(src, dst), score(ru, rv)), auc(cos(src, dst), cos(ru, rv))


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--nodes", type=int, default=1_000_000)
    ap.add_argument("--m", type=int, default=10)
    ap.add_argument("--walks", type=int, default=1_000_000)
    ap.add_argument("--epochs", type=int, default=1)
    ap.add_argument("--d", type=int, default=128)
    ap.add_argument("--lr", type=float, default=0.01)
    ap.add_argument("--modes", default="write_through,write_back,atomic")
    ap.add_argument("--hot-rows", type=int, default=None,
                    help="blocks modes: hot rows per cell (see bench.py; default the library's)")
    ap.add_argument("--hot-flush", type=int, default=0, help="blocks modes: see bench.py")
    ap.add_argument("--round-walks", type=int, default=1 << 19, help="blocks modes: walks per round")
    ap.add_argument("--stripes", type=int, default=1, help="blocks modes: centre stripes")
    ap.add_argument("--central-store", action="store_true",
                    help="blocks modes: single-run centres by a store instead of atomics")
    ap.add_argument("--group-parts", type=int, default=0, help="blocks modes: parts per group")
    ap.add_argument("--star", type=int, default=0,
                    help="join node 0 to this many random nodes: a hub that dominates its cell "
                         "(the hub-skew rule of gn2v_block_auto_plan_graph)")
    a = ap.parse_args()
    g = E.barabasi_albert(a.nodes, a.m, 42)
    if a.star:
        import numpy as np

        t = g._device_tensors
        rp, ci = t["row_ptr"].cpu().numpy().astype(np.int64), t["col_idx"].cpu().numpy().astype(np.int64)
        src = np.repeat(np.arange(a.nodes, dtype=np.int64), np.diff(rp))
        keep = src < ci
        far = np.random.RandomState(7).choice(np.arange(1, a.nodes), size=a.star, replace=False)
        g = E.CSRGraph.from_edge_list(np.concatenate([src[keep], np.zeros(a.star, dtype=np.int64)]),
                                      np.concatenate([ci[keep], far]), number_of_nodes=a.nodes)
        deg = np.diff(g.row_ptr)
        print(f"star of {a.star}: max degree {int(deg.max())}, directed edges {len(g.col_idx)}, "
              f"hub x 256 / edges = {deg.max() * 256 / len(g.col_idx):.2f}", flush=True)
    n, d = g.get_number_of_nodes(), a.d
    wp = ops.walk_params(128, 10, 0.25, 4.0)
    flagmap = {"write_through": _lib.TRAIN_WRITE_THROUGH, "write_back": _lib.TRAIN_WRITE_BACK,
               "atomic": _lib.TRAIN_ATOMIC}
    gen = torch.Generator(device="cuda")
    results = {}
    for mode in a.modes.split(","):
        if mode.startswith("blocks"):
            # blocks:<parts>:<slices>[:<record>] -- the block trainer on one GPU
            from embiggen_amd.distributed import BlockPartitionedTrainer, LoopbackComm

            f = mode.split(":")
            parts, slices = int(f[1]), int(f[2])
            record = int(f[3]) if len(f) > 3 and f[3] else 32
            # 5th field: "st" = plain stores on the XCD-exclusive contextual rows named explicitly,
            # so that they also apply below 2^16 nodes, where the automatic choice is atomics on
            # every row; "at" = atomics on every row
            kind = f[4] if len(f) > 4 else ""
            extra = {"": 0, "st": _lib.TRAIN_WRITE_THROUGH, "at": _lib.TRAIN_ATOMIC}[kind]
            if a.central_store:
                extra |= _lib.TRAIN_CENTRAL_STORE
            tp = ops.train_params(0, d, 10, 5, flags=1 | extra, ld=d)
            tr = BlockPartitionedTrainer(g, tp, d, d, 42, d ** -0.5, LoopbackComm(), "cuda:0",
                                         walk_length=128, window=5, parts=parts, slices=slices,
                                         record=record,
                                         hot_rows=a.hot_rows, hot_flush=a.hot_flush,
                                         stripes=a.stripes, group_parts=a.group_parts or None)
            ops.stats_reset(g)
            t0 = time.time()
            lr, rounds = a.lr, []
            for e in range(a.epochs):
                for first in range(0, a.walks, a.round_walks):
                    nb = min(a.round_walks, a.walks - first)
                    rounds.append((lambda e=e, first=first, nb=nb: ops.walks(g, wp, 42, e, first, nb),
                                   42, e, lr, first))
                lr *= 0.9
            tr.run(rounds, overlap=False)
            c, x = tr.gather_full()
            st = ops.stats_read(g)
            gen.manual_seed(1)
            res = evaluate(g, c, x, 200000, gen)
            print(f"{mode:14s} pairs={st['pairs']:.3e} train_ms={st['train_ms']:.0f} "
                  f"({st['pairs'] / st['train_ms'] * 1e3:.3e} pairs/s) wall={time.time() - t0:.1f}s "
                  f"AUC(c.x)={res[0]:.4f} AUC(cos central)={res[1]:.4f} "
                  f"finite={bool(torch.isfinite(c).all() and torch.isfinite(x).all())} "
                  f"|c|max={float(c.abs().max()):.3f} |x|max={float(x.abs().max()):.3f}", flush=True)
            del c, x, tr
            continue
        c = ops.init_table(n, d, 42, 0, d ** -0.5)
        x = ops.init_table(n, d, 42, 1, d ** -0.5)
        tp = ops.train_params(0, d, 10, 5, flags=1 | flagmap[mode])
        gen.manual_seed(1)
        base = evaluate(g, c, x, 200000, gen)
        ops.stats_reset(g)
        t0 = time.time()
        lr = a.lr
        for e in range(a.epochs):
            for first in range(0, a.walks, 1 << 16):
                nb = min(1 << 16, a.walks - first)
                wk = ops.walks(g, wp, 42, e, first, nb)
                ops.sgns_step(g, tp, wk, 42, e, first, lr, c, x)
            lr *= 0.9
        st = ops.stats_read(g)
        gen.manual_seed(1)
        res = evaluate(g, c, x, 200000, gen)
        results[mode] = res
        print(f"{mode:14s} pairs={st['pairs']:.3e} train_ms={st['train_ms']:.0f} "
              f"({st['pairs'] / st['train_ms'] * 1e3:.3e} pairs/s) wall={time.time() - t0:.1f}s "
              f"AUC(c.x)={res[0]:.4f} AUC(cos central)={res[1]:.4f}  [init: {base[0]:.4f} {base[1]:.4f}] "
              f"finite={bool(torch.isfinite(c).all() and torch.isfinite(x).all())} "
              f"|c|max={float(c.abs().max()):.3f} |x|max={float(x.abs().max()):.3f}", flush=True)
        del c, x

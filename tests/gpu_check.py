"""First-contact GPU check: parity of every kernel against the oracle + a quick speed probe.
Run on the GPU box:  python tests/gpu_check.py [--perf N_NODES]
"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

import embiggen_amd as E  # noqa: E402
from embiggen_amd import _lib, ops  # noqa: E402
from oracle import oracle as O  # noqa: E402


def check(name, ok, extra=""):
    print(f"[{'OK' if ok else 'FAIL'}] {name} {extra}", flush=True)
    return ok


def parity():
    good = True
    g = E.karate_club()
    og = O.OracleGraph(g.row_ptr, g.col_idx)
    # walks, first and second order
    for rw, ew in ((1.0, 1.0), (0.25, 4.0), (2.0, 0.5), (1e-3, 1e-3)):
        wp = ops.walk_params(32, 10, rw, ew)
        got = ops.walks(g, wp, 42, 3, 0, 340).cpu().numpy().view(np.uint32)
        ref = O.walks(og, O.WalkParams(32, 10, rw, ew, 100, 0), 42, 3, 0, 340)
        good &= check(f"walks rw={rw} ew={ew}", np.array_equal(got, ref),
                      f"mismatch={int((got != ref).sum())}")
    # BA edges
    src = torch.empty(999 * 3, dtype=torch.int32, device="cuda")
    dst = torch.empty_like(src)
    _lib.check(_lib.lib().gn2v_ba_edges(1000, 3, 42, src.data_ptr(), dst.data_ptr(), None))
    torch.cuda.synchronize()
    rs, rd = O.ba_edges(1000, 3, 42)
    good &= check("ba edges", np.array_equal(src.cpu().numpy().view(np.uint32), rs)
                  and np.array_equal(dst.cpu().numpy().view(np.uint32), rd))
    # init
    t = ops.init_table(34, 8, 42, 1, 0.35).cpu().numpy()
    good &= check("init table", np.array_equal(t, O.init_table(34, 8, 8, 42, 1, 0.35)))
    # one deterministic step, both models, d = 8 and d = 100
    for model, name in ((0, "sgns"), (1, "cbow")):
        for d in (8, 100, 128):
            ld = (d + 3) // 4 * 4
            wp = ops.walk_params(16, 2, 0.25, 4.0)
            wk = ops.walks(g, wp, 7, 0, 0, 68)
            wk_h = wk.cpu().numpy().view(np.uint32)
            for flags, label in ((1 | 8, "det"), (1, "write_through"), (1 | 32, "write_back"), (1 | 16, "atomic")):
                c = ops.init_table(34, d, 7, 0, d ** -0.5)
                x = ops.init_table(34, d, 7, 1, d ** -0.5)
                c_h, x_h = c.cpu().numpy().copy(), x.cpu().numpy().copy()
                tp = ops.train_params(model, d, 5, 3, flags=flags)
                otp = O.TrainParams(model, d, ld, 1, 5, 3, 0.01, 0.9, 6.0, 1, d ** -0.5)
                fn = ops.sgns_step if model == 0 else ops.cbow_step
                fn(g, tp, wk, 7, 0, 0, 0.05, c, x)
                torch.cuda.synchronize()
                O.train_walks(og, otp, wk_h, 7, 0, 0, 0.05, c_h, x_h)
                err = max(np.abs(c.cpu().numpy() - c_h).max(), np.abs(x.cpu().numpy() - x_h).max())
                tol = 1e-5 if label == "det" else 10.0  # parallel modes collide on 34 nodes: finite only
                good &= check(f"{name} step d={d} {label}", err < tol, f"err={err:.3e}")
    # full fit deterministic vs oracle
    for cls, mid in ((E.Node2VecSkipGramEnsmallen, 0), (E.Node2VecCBOWEnsmallen, 1)):
        m = cls(embedding_size=8, epochs=3, walk_length=16, iterations=2, window_size=3,
                number_of_negative_samples=4, verbose=False)
        m._model.deterministic = True
        t0 = time.time()
        res = m.fit_transform(g, return_dataframe=False).get_all_node_embedding()
        dt = time.time() - t0
        if mid == 1:
            res = list(reversed(res))
        rc, rx, pairs = O.fit(og, O.WalkParams(16, 2, 0.25, 4.0, 100, 0),
                              O.TrainParams(mid, 8, 8, 3, 4, 3, 0.01, 0.9, 6.0, 1, 8 ** -0.5), 42)
        err = max(np.abs(res[0] - rc).max(), np.abs(res[1] - rx).max())
        good &= check(f"fit det {cls.__name__}", err < 1e-4 and m.get_last_stats()["pairs"] == pairs,
                      f"err={err:.3e} pairs={pairs} t={dt:.2f}s")
    return good


def perf(n_nodes, m, n_walks, reps):
    t0 = time.time()
    g = E.barabasi_albert(n_nodes, m, 42)
    torch.cuda.synchronize()
    print(f"BA graph {n_nodes} nodes, {g.get_number_of_directed_edges()} directed edges "
          f"built in {time.time() - t0:.2f}s", flush=True)
    wp = ops.walk_params(128, 10, 0.25, 4.0)
    for label, wpp in (("p=q=1", ops.walk_params(128, 10, 1.0, 1.0)), ("rw.25/ew4", wp),
                       ("rw2/ew.5", ops.walk_params(128, 10, 2.0, 0.5))):
        ops.stats_reset(g)
        wk = ops.walks(g, wpp, 42, 0, 0, n_walks)
        st = ops.stats_read(g)
        print(f"walks {label}: {st['walk_steps']} steps in {st['walk_ms']:.2f} ms -> "
              f"{st['walk_steps'] / st['walk_ms'] * 1e3:.3e} steps/s", flush=True)
    d = 128
    for flags, label in ((1, "write_through"), (1 | 32, "write_back"), (1 | 16, "atomic")):
        c = ops.init_table(n_nodes, d, 42, 0, d ** -0.5)
        x = ops.init_table(n_nodes, d, 42, 1, d ** -0.5)
        tp = ops.train_params(0, d, 10, 5, flags=flags)
        for r in range(reps):
            ops.stats_reset(g)
            ops.sgns_step(g, tp, wk, 42, 0, 0, 0.01, c, x)
            st = ops.stats_read(g)
            gbs = st["pairs"] * 12288 / (st["train_ms"] * 1e-3) / 1e9
            print(f"sgns {label} rep{r}: {st['pairs']} pairs in {st['train_ms']:.1f} ms -> "
                  f"{st['pairs'] / st['train_ms'] * 1e3:.3e} pairs/s, {gbs:.0f} GB/s algorithmic "
                  f"({gbs / 8000:.3f} of 8 TB/s)", flush=True)
        print("finite:", bool(torch.isfinite(c).all().item() and torch.isfinite(x).all().item()))
        del c, x


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--perf", type=int, default=0)
    ap.add_argument("--m", type=int, default=10)
    ap.add_argument("--walks", type=int, default=1 << 16)
    ap.add_argument("--reps", type=int, default=2)
    ap.add_argument("--skip-parity", action="store_true")
    a = ap.parse_args()
    ok = True
    if not a.skip_parity:
        ok = parity()
    if a.perf:
        perf(a.perf, a.m, a.walks, a.reps)
    sys.exit(0 if ok else 1)

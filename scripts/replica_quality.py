"""Quality of replica data-parallel training (simulated on one GPU): W replicas train on disjoint
walk slices per step and are merged by delta-sum or delta-mean; compared with one trainer."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import embiggen_amd as E
from embiggen_amd import ops, _lib
from sharded_helpers import link_auc_device as _auc
nodes, total, per_step = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
lr = 0.025
g = E.barabasi_albert(nodes, 8, 42); n = g.get_number_of_nodes(); d = 64
wp = ops.walk_params(64, 1, 1.0, 1.0)
tp = ops.train_params(0, d, 5, 4, flags=1)
gen = torch.Generator(device="cuda")
def fresh(): return ops.init_table(n, d, 42, 0, d ** -0.5), ops.init_table(n, d, 42, 1, d ** -0.5)
c, x = fresh()
for first in range(0, total, per_step):
    ops.sgns_step(g, tp, ops.walks(g, wp, 42, 0, first, per_step), 42, 0, first, lr, c, x)
gen.manual_seed(1); print("single", round(_auc(g, c, x, gen), 4), flush=True)
deg = torch.from_numpy(g.get_node_degrees()).cuda().float()
hot_frac = float(sys.argv[4]) if len(sys.argv) > 4 else 0.01
hot = torch.zeros(n, 1, device="cuda")
hot[torch.argsort(deg, descending=True)[: int(n * hot_frac)]] = 1.0
for W in (2, 8):
    for merge in ("sum", "hybrid", "hybrid-touch"):
        bc, bx = fresh()
        for step in range(total // (per_step * W)):
            acc_c, acc_x = torch.zeros_like(bc), torch.zeros_like(bx)
            cnt_c, cnt_x = torch.zeros(n, 1, device="cuda"), torch.zeros(n, 1, device="cuda")
            for r in range(W):
                rc, rx = bc.clone(), bx.clone()
                first = (step * W + r) * per_step
                ops.sgns_step(g, tp, ops.walks(g, wp, 42, 0, first, per_step), 42, 0, first, lr, rc, rx)
                dc, dx = rc - bc, rx - bx
                acc_c += dc; acc_x += dx
                cnt_c += (dc.abs().amax(1, keepdim=True) > 0).float()
                cnt_x += (dx.abs().amax(1, keepdim=True) > 0).float()
            if merge == "sum":
                bc += acc_c; bx += acc_x
            elif merge == "hybrid":  # hot rows averaged, the rest summed
                bc += acc_c * (1 - hot) + acc_c * hot / W; bx += acc_x * (1 - hot) + acc_x * hot / W
            elif merge == "hybrid-touch":  # hot rows: mean over the ranks that touched them
                bc += acc_c * (1 - hot) + acc_c * hot / cnt_c.clamp_min(1)
                bx += acc_x * (1 - hot) + acc_x * hot / cnt_x.clamp_min(1)
            elif merge == "mean":
                bc += acc_c / W; bx += acc_x / W
            else:
                bc += acc_c / cnt_c.clamp_min(1); bx += acc_x / cnt_x.clamp_min(1)
        gen.manual_seed(1); print(f"W={W} merge={merge} per_step={per_step}", round(_auc(g, bc, bx, gen), 4), flush=True)

"""How evenly the pairs of a round spread over the 8 XCD cells of a part (the launch of a part
ends when its fullest cell is done):  python scripts/cell_balance.py [nodes] [round_walks]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import embiggen_amd as E  # noqa: E402
from embiggen_amd import ops  # noqa: E402
from embiggen_amd.distributed import BlockPartitionedTrainer, LoopbackComm  # noqa: E402

nodes = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
walks = int(sys.argv[2]) if len(sys.argv) > 2 else 1 << 21
g = E.barabasi_albert(nodes, 10, 42)
tp = ops.train_params(0, 128, 10, 5, flags=1, ld=128)
tr = BlockPartitionedTrainer(g, tp, 128, 128, 42, 128 ** -0.5, LoopbackComm(), "cuda:0",
                             walk_length=128, window=5)
wp = ops.walk_params(128, 10, 0.25, 4.0)
wk = ops.walks(g, wp, 42, 0, 0, walks)
# (a trainer of resident cells extracts under the round's placement: round_state)
_, offsets, n, _, _, _ = tr.prepare(wk, 42, 0, 0, rstate=tr.round_state(wk, 42, 0))
sizes = (offsets[1:] - offsets[:-1]).to(torch.float64).reshape(tr.parts, tr.slices)
ratio = sizes.max(dim=1).values / sizes.mean(dim=1)
print(f"{nodes} nodes, {walks} walks: {n} pairs, {tr.parts} parts x {tr.slices} slices; fullest "
      f"cell / mean cell of a part: mean {float(ratio.mean()):.4f}, max {float(ratio.max()):.4f}; "
      f"parts max / mean {float(sizes.sum(1).max() / sizes.sum(1).mean()):.4f}")

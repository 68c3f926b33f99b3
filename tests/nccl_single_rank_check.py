"""Run by tests/test_gpu_blocks.py in a child process: one-rank RCCL ("nccl") process group,
block-partitioned trainer through TorchComm vs LoopbackComm.  Prints OK <max abs diff>."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import embiggen_amd as E  # noqa: E402
from embiggen_amd import _lib, ops  # noqa: E402
from embiggen_amd.distributed import BlockPartitionedTrainer, LoopbackComm, TorchComm  # noqa: E402

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29541")
torch.cuda.set_device(0)
dist.init_process_group(backend="nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))

g = E.karate_club()
D, K = 16, 4
tp = ops.train_params(0, D, K, 3, flags=1 | _lib.TRAIN_DETERMINISTIC)
wp = ops.walk_params(20, 2, 0.25, 4.0)


def run(comm):
    tr = BlockPartitionedTrainer(g, tp, D, D, 42, D ** -0.5, comm, "cuda:0", walk_length=20,
                                 window=3, parts=2)
    for r in range(2):
        walks = ops.walks(g, wp, 42, 0, r * 34, 34)
        tr.train_round(walks, 42, 0, 0.02, r * 34)
    return [t.cpu().numpy() for t in tr.gather_full()], tr.last_round


comm = TorchComm()
assert comm.backend == "nccl" and comm.world == 1
(a0, a1), info = run(comm)
(b0, b1), _ = run(LoopbackComm())
# raw collectives on device tensors, the shapes the trainer uses for world > 1
walks = torch.arange(40, dtype=torch.int32, device="cuda").reshape(2, 20)
assert torch.equal(comm.all_gather(walks), walks)
rows = torch.arange(12, dtype=torch.float32, device="cuda").reshape(4, 3)
back = torch.empty_like(rows)
comm.sendrecv_start(rows, 0, back, 0).wait()
torch.cuda.synchronize()
assert torch.equal(back, rows)
dist.barrier()
dist.destroy_process_group()
diff = max(np.abs(a0 - b0).max(), np.abs(a1 - b1).max())
assert info["pairs_trained"] > 0 and np.isfinite(a0).all()
print("OK", diff)

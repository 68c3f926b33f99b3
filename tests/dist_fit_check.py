"""Run by tests/test_gpu_blocks.py under ``python -m torch.distributed.run`` with 2 ranks sharing
GPU 0 over gloo: the public model's ``fit_transform`` with an explicit ``comm`` (the opt-in
multi-GPU path) -- every rank gets the same full tables; rank 0 writes them for the caller."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import embiggen_amd as E  # noqa: E402
from embiggen_amd.distributed import TorchComm  # noqa: E402
from helpers import link_auc, ring_of_cliques  # noqa: E402

out_dir = sys.argv[1]
torch.cuda.set_device(0)
dist.init_process_group(backend="gloo")
rank = dist.get_rank()
src, dst, n = ring_of_cliques(32, 8)
g = E.CSRGraph.from_edge_list(src, dst, number_of_nodes=n)
model = E.Node2VecSkipGramEnsmallen(embedding_size=16, epochs=3, walk_length=32, iterations=4,
                                    window_size=4, number_of_negative_samples=5,
                                    learning_rate=0.025, return_weight=1.0, explore_weight=1.0,
                                    verbose=False)
model.set_distributed(TorchComm())
res = model.fit_transform(g, return_dataframe=False).get_all_node_embedding()
np.save(os.path.join(out_dir, f"central{rank}.npy"), res[0])
np.save(os.path.join(out_dir, f"contextual{rank}.npy"), res[1])
# a model without a multi-GPU path inside the same job runs on its own device
cbow = E.Node2VecCBOWEnsmallen(embedding_size=8, epochs=1, walk_length=8, iterations=1,
                               window_size=2, verbose=False)
cbow.set_distributed(model._model.comm)
assert cbow.fit_transform(g, return_dataframe=False).get_all_node_embedding()[0].shape == (n, 8)
dist.barrier()
if rank == 0:
    print("OK", model._model.last_plan, round(link_auc(g, res[0], res[1]), 4), flush=True)
dist.destroy_process_group()

"""One-process-per-GPU data parallelism for the training path (torch.distributed; backend
"nccl" = RCCL over xGMI on the GPU box, "gloo" in the CPU tests).

Round-1 form: the CSR and both tables are replicated, walks are partitioned
by walk id, and after every step the replicas exchange what they learned.  Because every rank only
touches the rows its walks visit, replicas are combined by **summing deltas**
    new = base + sum_r (table_r - base)  =  all_reduce_sum(table_r) - (world - 1) * base
rather than by averaging (averaging would divide the update of a row only one rank touched by the
world size).  One all-reduce per table per step: 2 x 4*N*ld bytes per 1250 * walks_per_step pairs.
The reference has no counterpart: ensmallen parallelises with rayon threads inside one process
(SURVEY.md section 2a).
"""
from typing import Tuple


def walk_slice(step: int, rank: int, world: int, walks_per_step: int) -> Tuple[int, int]:
    """(first_walk_id, n_walks) trained by `rank` in global step `step`: slices are disjoint
    across ranks and steps and cover the walk ids contiguously."""
    return (step * world + rank) * walks_per_step, walks_per_step


class ReplicaSync:
    """Keeps the last agreed copy of each table and folds every rank's delta into it."""

    def __init__(self, *tables):
        import torch.distributed as dist

        self._dist = dist
        self.world = dist.get_world_size() if dist.is_initialized() else 1
        self.tables = tables
        self.bases = [t.clone() for t in tables] if self.world > 1 else []

    def sync(self):
        """Blocking on the current stream: after it every rank holds base + sum of all deltas."""
        if self.world == 1:
            return
        for table, base in zip(self.tables, self.bases):
            self._dist.all_reduce(table, op=self._dist.ReduceOp.SUM)
            table.add_(base, alpha=-(self.world - 1))
            base.copy_(table)

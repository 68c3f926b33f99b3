"""Test harness for the row-sharded trainer: an in-process communicator (threads + barrier) that
behaves like an all-to-all fabric, and an oracle-backed compute callable (tests only)."""
import threading

import numpy as np
import torch

from oracle import oracle as O


class ThreadComm:
    """W ranks as W threads of one process; exchange through shared slots."""

    backend = "threads"

    class Shared:
        def __init__(self, world):
            self.world = world
            self.barrier = threading.Barrier(world)
            self.slots = [None] * world
            self.mail = {}

    def __init__(self, shared, rank):
        self.shared, self.rank, self.world = shared, rank, shared.world

    def all_gather(self, tensor):
        s = self.shared
        if tensor.is_cuda:  # the other "ranks" read it from their own streams
            torch.cuda.current_stream(tensor.device).synchronize()
        s.slots[self.rank] = tensor
        s.barrier.wait()
        out = torch.cat([s.slots[r] for r in range(self.world)]).clone()
        s.barrier.wait()
        return out

    def sendrecv_start(self, send, dst, recv, src):
        s, rank = self.shared, self.rank
        s.mail[(rank, dst)] = send.clone()
        if send.is_cuda:
            torch.cuda.current_stream(send.device).synchronize()

        class Handle:
            def wait(self_inner):
                s.barrier.wait()
                recv.copy_(s.mail[(src, rank)])
                s.barrier.wait()

        return Handle()

    def broadcast(self, tensor, src):
        s = self.shared
        if tensor.is_cuda:
            torch.cuda.current_stream(tensor.device).synchronize()
        if self.rank == src:
            s.mail[("bcast", src)] = tensor.clone()
        s.barrier.wait()
        tensor.copy_(s.mail[("bcast", src)])
        s.barrier.wait()
        return tensor

    def send_to_root(self, tensor, root, buffer=None, src=None):
        s = self.shared
        if src == root:
            return tensor if self.rank == root else None
        if self.rank == src:
            if tensor.is_cuda:
                torch.cuda.current_stream(tensor.device).synchronize()
            s.mail[("root", src)] = tensor.clone()
        s.barrier.wait()
        out = None
        if self.rank == root:
            buffer.copy_(s.mail[("root", src)])
            out = buffer
        s.barrier.wait()
        return out


def run_ranks(world, fn):
    """Run fn(comm) for every rank in its own thread; returns the list of results."""
    shared = ThreadComm.Shared(world)
    results, errors = [None] * world, []

    def target(rank):
        try:
            results[rank] = fn(ThreadComm(shared, rank))
        except BaseException as e:  # noqa: BLE001
            errors.append(e)
            shared.barrier.abort()

    threads = [threading.Thread(target=target, args=(r,)) for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    if errors:
        raise errors[0]
    return results


class OracleBlockBackend:
    """CPU stand-in for embiggen_amd.distributed.GpuBlockBackend: the same five operations done
    by the oracle on numpy arrays behind CPU torch tensors (tests only)."""

    def __init__(self, graph):
        self.graph = graph
        self.og = O.OracleGraph(graph.row_ptr, graph.col_idx)

    def init_rows(self, n_rows, d, ld, seed, table_id, scale, first_row, stride, out=None):
        t = torch.from_numpy(O.init_table_rows(n_rows, d, ld, seed, table_id, scale, first_row,
                                               stride))
        if out is None:
            return t
        out.copy_(t)
        return out

    def empty_rows(self, n_rows, ld):
        return torch.zeros((n_rows, ld), dtype=torch.float32)

    def plan(self, world, rank, parts, slices, walk_length, window, min_dist, record, flags,
             hot_rows=0, hot_flush=0):
        return O.block_plan(self.graph.get_number_of_nodes(), world, rank, parts, slices,
                            walk_length, window, min_dist, record, flags, hot_rows)

    def alias_tables(self, plan, inv=None, out=None):
        return O.block_alias(self.og, plan.parts, plan.slices, plan.hot_rows, inv=inv)

    def placement(self, classes, seed, round_id, out=None):
        return O.block_placement(self.graph.get_number_of_nodes(), classes, seed, round_id)

    def place_walks(self, place, walks_all):
        # o_block_extract maps the node ids itself: what `prepare` receives as "the placed
        # walks" is the placement
        return place

    def prepare(self, plan, walks_all, seed, epoch, first_walk, hub_bits=None, part_lo=0,
                part_n=0, placed=None):
        walks = walks_all.cpu().numpy().view(np.uint32)
        words, offsets = O.block_extract(self.og, plan, walks, seed, epoch, first_walk,
                                         hub_bits=hub_bits, part_lo=part_lo, part_n=part_n,
                                         place=placed)
        return words, offsets, len(words)

    def step(self, tp, plan, prepared, alias, cell_rows, central, context, block_id, part, seed,
             epoch, lr, whole_central=False, hot=None, inv=None, context_table=None):
        words, offsets, n_pairs = prepared[:3]
        if n_pairs == 0:
            return
        # a centre stripe of the whole table is handed to the oracle as the partition it is
        mine = central[plan.rank::plan.world] if whole_central else central
        c = np.ascontiguousarray(mine.numpy())
        rows = context_table if context_table is not None else context
        x = np.ascontiguousarray(rows.numpy())
        O.block_step(self.og, tp, plan, words, offsets, alias, cell_rows, c, x, block_id,
                     part, seed, epoch, lr, inv=inv, natural=context_table is not None)
        mine.copy_(torch.from_numpy(c))
        rows.copy_(torch.from_numpy(x))


def link_auc_device(g, c, x, gen, n_eval=100000):
    """AUROC of the symmetrised SkipGram score for sampled edges vs random node pairs, on the GPU
    (graph must be device resident)."""
    t = g._device_tensors
    n = g.get_number_of_nodes()
    e = torch.randint(0, t["col_idx"].numel(), (n_eval,), device="cuda", generator=gen)
    dst = t["col_idx"][e].long()
    src = torch.searchsorted(t["row_ptr"], e, right=True) - 1
    ru = torch.randint(0, n, (n_eval,), device="cuda", generator=gen)
    rv = torch.randint(0, n, (n_eval,), device="cuda", generator=gen)
    score = lambda u, v: (c[u] * x[v]).sum(1) + (c[v] * x[u]).sum(1)  # noqa: E731
    s = torch.cat([score(src, dst), score(ru, rv)])
    ranks = torch.empty_like(s)
    ranks[torch.argsort(s)] = torch.arange(1, s.numel() + 1, device="cuda", dtype=s.dtype)
    return float((ranks[:n_eval].sum() - n_eval * (n_eval + 1) / 2) / (n_eval * n_eval))

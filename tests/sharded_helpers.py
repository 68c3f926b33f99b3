"""Test harness for the row-sharded trainer: an in-process communicator (threads + barrier) that
behaves like an all-to-all fabric, and an oracle-backed compute callable (tests only)."""
import threading

import numpy as np
import torch

from oracle import oracle as O


class ThreadComm:
    """W ranks as W threads of one process; exchange through shared slots."""

    class Shared:
        def __init__(self, world):
            self.world = world
            self.barrier = threading.Barrier(world)
            self.slots = [None] * world

    def __init__(self, shared, rank):
        self.shared, self.rank, self.world = shared, rank, shared.world

    def exchange_counts(self, counts):
        s = self.shared
        s.slots[self.rank] = counts
        s.barrier.wait()
        out = torch.stack([s.slots[r][self.rank] for r in range(self.world)])
        s.barrier.wait()
        return out

    def exchange_rows(self, rows, send_counts, recv_counts):
        s = self.shared
        s.slots[self.rank] = (rows, list(send_counts))
        s.barrier.wait()
        pieces = []
        for r in range(self.world):
            src, counts = s.slots[r]
            start = sum(counts[: self.rank])
            pieces.append(src[start:start + counts[self.rank]])
            assert counts[self.rank] == recv_counts[r]
        out = torch.cat(pieces).clone()
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


def host_init_fn(n_nodes, d, ld, seed, scale):
    def init_fn(table_id):
        return torch.from_numpy(O.init_table(n_nodes, d, ld, seed, table_id, scale))

    return init_fn


def oracle_block_compute(oracle_graph, otp, trainer):
    """compute callable for BlockPartitionedTrainer backed by the oracle's pair-mode step."""

    def compute(pairs, rows, part, seed, epoch, first_pair, lr):
        host = [np.ascontiguousarray(t.detach().cpu().numpy())
                for t in (trainer.central, trainer.context)]
        tp = type(otp).from_buffer_copy(otp)
        tp.window = pairs.shape[1] - 1  # (centre, context) pairs or centre records
        O.train_walks_ex(
            oracle_graph, tp, pairs.cpu().numpy().view(np.uint32), seed, epoch, first_pair, lr,
            host[0], host[1], walk_rows=rows.cpu().numpy().view(np.uint32),
            neg_pool=trainer.pools[part].cpu().numpy().view(np.uint32),
            neg_id_mul=trainer.comm.world, neg_id_add=part, pair_mode=True)
        trainer.central.copy_(torch.from_numpy(host[0]))
        trainer.context.copy_(torch.from_numpy(host[1]))

    return compute


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


def host_walk_pair_blocks(walks_tensor, window, min_dist, world, salt):
    """CPU stand-in for ops.walk_pair_blocks in its grouping mode (salt = 2^64 - 1): the same
    slots and keys (block << 32 | centre, INT64_MAX for unused slots), built from the oracle's
    co-occurrence slots (tests only)."""
    assert salt == 2 ** 64 - 1
    walks = walks_tensor.cpu().numpy().view(np.uint32)
    keys, _ = O.cooc_slots(walks, window, min_dist)
    used = keys != O.COOC_UNUSED
    c, x = (keys >> np.uint64(32)).astype(np.int64), (keys & np.uint64(0xFFFFFFFF)).astype(np.int64)
    slots = np.full((len(keys), 2), -1, dtype=np.int32)
    slots[used, 0] = c[used].astype(np.uint32).view(np.int32)
    slots[used, 1] = x[used].astype(np.uint32).view(np.int32)
    out = np.full(len(keys), 0x7FFFFFFFFFFFFFFF, dtype=np.int64)
    out[used] = (((c[used] % world) * world + x[used] % world) << 32) | c[used]
    return torch.from_numpy(slots), torch.from_numpy(out)

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


def oracle_compute(oracle_graph, otp, tables):
    """compute callable for ShardedTrainer backed by the oracle's general step (works for CPU
    tensors in place and for CUDA tensors through a host round trip)."""

    def compute(walks, rows, cache_c, cache_x, seed, epoch, first_walk, lr):
        neg_t = tables.central if otp.model == 1 else tables.contextual
        host = [t.detach().cpu().numpy() for t in (cache_c, cache_x, neg_t)]
        host = [np.ascontiguousarray(h) for h in host]
        O.train_walks_ex(
            oracle_graph, otp, walks.cpu().numpy().view(np.uint32), seed, epoch, first_walk, lr,
            host[0], host[1], walk_rows=rows.cpu().numpy().view(np.uint32), negative=host[2],
            neg_pool=tables.neg_pool.cpu().numpy().view(np.uint32),
            neg_id_mul=tables.comm.world, neg_id_add=tables.comm.rank)
        for t, h in zip((cache_c, cache_x, neg_t), host):
            t.copy_(torch.from_numpy(h))

    return compute


def host_init_fn(n_nodes, d, ld, seed, scale):
    def init_fn(table_id):
        return torch.from_numpy(O.init_table(n_nodes, d, ld, seed, table_id, scale))

    return init_fn


def oracle_block_compute(oracle_graph, otp, trainer):
    """compute callable for BlockPartitionedTrainer backed by the oracle's pair-mode step."""

    def compute(pairs, rows, part, seed, epoch, first_pair, lr):
        host = [np.ascontiguousarray(t.detach().cpu().numpy())
                for t in (trainer.central, trainer.context)]
        O.train_walks_ex(
            oracle_graph, otp, pairs.cpu().numpy().view(np.uint32), seed, epoch, first_pair, lr,
            host[0], host[1], walk_rows=rows.cpu().numpy().view(np.uint32),
            neg_pool=trainer.pools[part].cpu().numpy().view(np.uint32),
            neg_id_mul=trainer.comm.world, neg_id_add=part, pair_mode=True)
        trainer.central.copy_(torch.from_numpy(host[0]))
        trainer.context.copy_(torch.from_numpy(host[1]))

    return compute

"""Batch generator of (contexts, words) windows from second-order random walks.

Counterpart of the reference's Keras sequence
(embiggen/sequences/tensorflow_sequences/node2vec_sequence.py:11-203): same constructor arguments
and defaults (:14-27), same batch contract -- item ``idx`` is ``(((contexts, words),),)`` with
``contexts`` int32 ``[n, 2*window_size]``, ``words`` int32 ``[n]`` and
``n = batch_size * iterations * (walk_length - 2*window_size)`` (:115-128) -- and the same seeding
rule ``random_state + idx + elapsed_epochs`` (:200).  The reference delegates the batch to
``graph.node2vec(...)`` inside the ensmallen wheel (:190-201); here the walks come from
``gn2v_walks`` and the windows from ``gn2v_window_batch`` on the GPU.  It does not depend on Keras:
any training loop can index it or iterate ``__call__`` like the reference's generator (:102-105).
"""
from typing import Optional

import numpy as np

from .. import ops
from ..graph import CSRGraph


class Node2VecSequence:
    def __init__(
        self,
        graph: CSRGraph,
        walk_length: int = 128,
        batch_size: int = 256,
        iterations: int = 16,
        window_size: int = 4,
        return_weight: float = 1.0,
        explore_weight: float = 1.0,
        change_node_type_weight: float = 1.0,
        change_edge_type_weight: float = 1.0,
        max_neighbours: Optional[int] = 100,
        random_state: int = 42,
        device: int = 0,
        return_device_tensors: bool = False,
    ):
        if walk_length <= 2 * window_size:
            raise ValueError("walk_length must exceed 2 * window_size.")
        if not (change_node_type_weight > 0 and change_edge_type_weight > 0):
            raise ValueError(
                "change_node_type_weight and change_edge_type_weight must be strictly positive.")
        self._graph = graph
        self._walk_length = walk_length
        self._batch_size = batch_size
        self._iterations = iterations
        self._window_size = window_size
        self._return_weight = return_weight
        self._explore_weight = explore_weight
        self._change_node_type_weight = change_node_type_weight
        self._change_edge_type_weight = change_edge_type_weight
        self._max_neighbours = max_neighbours
        self._random_state = random_state
        self._device = device
        self._return_device_tensors = return_device_tensors
        self._current_index = 0
        self.elapsed_epochs = 0
        self._sample_number = graph.get_number_of_unique_source_nodes()

    # ------------------------------------------------------------------ Sequence protocol
    @property
    def sample_number(self) -> int:
        return self._sample_number

    @property
    def batch_size(self) -> int:
        return self._batch_size

    @property
    def steps_per_epoch(self) -> int:
        return max(int(np.ceil(self._sample_number / self._batch_size)), 1)

    def __len__(self) -> int:
        return self.steps_per_epoch

    def on_epoch_end(self):
        self.elapsed_epochs += 1

    def reset(self):
        self.elapsed_epochs = 0
        self._current_index = 0

    def __call__(self):
        self._current_index += 1
        return self[self._current_index]

    @property
    def number_of_skipgrams(self) -> int:
        return self._batch_size * self._iterations * (self._walk_length - 2 * self._window_size)

    def walks(self, idx: int):
        """The batch's walks on the device: int32 [batch_size * iterations, walk_length]."""
        import torch

        wp = ops.walk_params(self._walk_length, self._iterations, self._return_weight,
                             self._explore_weight, self._max_neighbours,
                             self._change_node_type_weight, self._change_edge_type_weight)
        seed = self._random_state + idx + self.elapsed_epochs
        first_source = (idx * self._batch_size) % self._sample_number
        # the whole batch in ONE launch (gn2v_walks_strided): walk b = source b % batch_size in
        # iteration b // batch_size, its id 2 * iteration * n_sources + first_source + b % batch_size
        # (ids spaced by 2 * n_sources per iteration keep wrapped-around sources distinct) -- a
        # walk is 127 dependent steps, so sixteen launches of 256 walks took sixteen times as long
        import ctypes as C

        from .. import _lib

        dev = torch.device("cuda", self._device)
        out = torch.empty((self._iterations * self._batch_size, self._walk_length),
                          dtype=torch.int32, device=dev)
        _lib.check(_lib.lib().gn2v_walks_strided(
            self._graph.device_graph(self._device).handle, C.byref(wp), seed, 0, first_source,
            self._iterations * self._batch_size, self._batch_size, 2 * self._sample_number,
            out.data_ptr(), torch.cuda.current_stream(dev).cuda_stream))
        return out

    def __getitem__(self, idx: int):
        contexts, words = ops.window_batch(self.walks(idx), self._window_size)
        if not self._return_device_tensors:
            contexts, words = contexts.cpu().numpy(), words.cpu().numpy()
        return (((contexts, words),),)

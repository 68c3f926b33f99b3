"""Engine-side model objects: the stand-ins for ``ensmallen.models.SkipGram`` / ``.CBOW``.

The reference constructs ``models.SkipGram(embedding_size=..., random_state=..., **kwargs)``
(embiggen/embedders/ensmallen_embedders/node2vec.py:65-69) and calls ``.fit_transform(graph)``
(:99), receiving ``[central, contextual]`` float32 ``[N, d]`` numpy arrays.  These classes keep
that interface and run the whole call on the GPU through ``gn2v_train`` (include/gn2v.h).
PyTorch only allocates the two tables in HBM and provides the stream.
"""
import ctypes as C
import os
import sys
import time
from typing import List, Optional

import numpy as np

from . import _lib
from .graph import CSRGraph

# `dtype` argument of the reference's wrappers (node2vec_skipgram.py:32,103-104) -> torch dtype name
_RESULT_DTYPES = {"f16": "float16", "f32": "float32", "f64": "float64"}


def _as_csr(graph, normalize_by_degree: bool = False) -> CSRGraph:
    if normalize_by_degree:
        return _as_csr(graph).with_degree_normalized_weights()
    if isinstance(graph, CSRGraph):
        return graph
    if hasattr(graph, "get_cumulative_node_degrees"):
        return CSRGraph.from_ensmallen(graph)
    raise TypeError(
        "The graph must be an embiggen_amd.CSRGraph (or an ensmallen.Graph, converted through "
        f"its CSR getters); got {type(graph)}."
    )


class _WalkBasedModel:
    MODEL_ID = None
    NAME = None

    def __init__(
        self,
        embedding_size: int = 100,
        random_state: int = 42,
        epochs: int = 30,
        clipping_value: float = 6.0,
        number_of_negative_samples: int = 10,
        walk_length: int = 128,
        iterations: int = 10,
        window_size: int = 5,
        return_weight: float = 0.25,
        explore_weight: float = 4.0,
        change_node_type_weight: float = 1.0,
        change_edge_type_weight: float = 1.0,
        max_neighbours: Optional[int] = 100,
        learning_rate: float = 0.01,
        learning_rate_decay: float = 0.9,
        central_nodes_embedding_path: Optional[str] = None,
        contextual_nodes_embedding_path: Optional[str] = None,
        normalize_by_degree: bool = False,
        stochastic_downsample_by_degree: bool = False,
        normalize_learning_rate_by_degree: bool = False,
        use_scale_free_distribution: bool = True,
        dtype: str = "f32",
        verbose: bool = True,
        alpha: float = 0.75,
        deterministic: bool = False,
        update_mode: str = "auto",
        device: int = 0,
        min_distance: int = 1,
        block_path: Optional[bool] = None,
    ):
        if not isinstance(embedding_size, int) or embedding_size < 1:
            raise ValueError("The embedding size must be a strictly positive integer.")
        if embedding_size > 1024:
            raise ValueError("Embedding sizes above 1024 are not supported by the gn2v engine.")
        if epochs < 0 or walk_length < 2 or iterations < 1 or window_size < 1:
            raise ValueError(
                "epochs must be >= 0, walk_length >= 2, iterations >= 1 and window_size >= 1."
            )
        if number_of_negative_samples < 0:
            raise ValueError("number_of_negative_samples must be >= 0.")
        if not (return_weight > 0 and explore_weight > 0):
            raise ValueError("return_weight and explore_weight must be strictly positive.")
        if not clipping_value > 0:
            raise ValueError("clipping_value must be strictly positive.")
        if dtype not in _RESULT_DTYPES:
            raise ValueError(
                f"dtype must be one of {sorted(_RESULT_DTYPES)}, got {dtype!r} (the engine always "
                "computes in f32; 'f16' / 'f64' convert the returned tables).")
        self.dtype = dtype
        if not (change_node_type_weight > 0 and change_edge_type_weight > 0):
            raise ValueError(
                "change_node_type_weight and change_edge_type_weight must be strictly positive.")
        self.embedding_size = embedding_size
        self.random_state = int(random_state)
        self.epochs = epochs
        self.clipping_value = float(clipping_value)
        self.number_of_negative_samples = number_of_negative_samples
        self.walk_length = walk_length
        self.iterations = iterations
        self.window_size = window_size
        self.return_weight = float(return_weight)
        self.explore_weight = float(explore_weight)
        # act on graphs that carry node / edge types, no impact otherwise
        # (node2vec_sequence.py:57-66)
        self.change_node_type_weight = float(change_node_type_weight)
        self.change_edge_type_weight = float(change_edge_type_weight)
        self.max_neighbours = max_neighbours
        self.learning_rate = float(learning_rate)
        self.learning_rate_decay = float(learning_rate_decay)
        self.central_nodes_embedding_path = central_nodes_embedding_path
        self.contextual_nodes_embedding_path = contextual_nodes_embedding_path
        self.normalize_by_degree = bool(normalize_by_degree)
        self.stochastic_downsample_by_degree = bool(stochastic_downsample_by_degree)
        self.normalize_learning_rate_by_degree = bool(normalize_learning_rate_by_degree)
        self.use_scale_free_distribution = bool(use_scale_free_distribution)
        self.verbose = bool(verbose)
        self.alpha = alpha
        self.deterministic = bool(deterministic)
        if update_mode not in ("auto", "write_through", "write_back", "atomic"):
            raise ValueError(
                "update_mode must be 'auto', 'write_through', 'write_back' or 'atomic'.")
        self.update_mode = update_mode
        self.device = int(device)
        if not 1 <= min_distance <= window_size:
            raise ValueError("min_distance must be in [1, window_size].")
        self.min_distance = int(min_distance)
        # None: the engine decides (SkipGram, >= 2 560 nodes, default update mode: block path);
        # True / False force or forbid it
        self.block_path = block_path
        self.last_plan = None
        self.last_stats = None
        self.last_seconds = None

    # ------------------------------------------------------------------ C structs
    @property
    def padded_size(self) -> int:
        """Row stride of the device tables: rows wider than 64 B are padded to whole 128 B cache
        lines (measured on the 10 M-node graph: d = 100 trains 13 % faster at stride 128 than at
        100-112), narrow rows only to 16 B; rows wider than 512 B to whole 256 B -- the gradient
        of a central row is added 256 contiguous bytes per atomic instruction, and an instruction
        that starts in the middle of such a window touches three cache lines instead of two
        (d = 200: 8.1e8 pairs/s at stride 224, 1.06e9 at 256)."""
        d = self.embedding_size
        if d <= 16:
            return (d + 3) // 4 * 4
        return (d + 31) // 32 * 32 if d <= 128 else (d + 63) // 64 * 64

    def walk_params(self) -> _lib.WalkParams:
        return _lib.WalkParams(
            self.walk_length, self.iterations, self.return_weight, self.explore_weight,
            0 if self.max_neighbours is None else int(self.max_neighbours), 0,
            self.change_node_type_weight, self.change_edge_type_weight,
        )

    def train_params(self) -> _lib.TrainParams:
        flags = 0
        if self.use_scale_free_distribution:
            flags |= _lib.TRAIN_SCALE_FREE
        if self.stochastic_downsample_by_degree:
            flags |= _lib.TRAIN_DOWNSAMPLE
        if self.normalize_learning_rate_by_degree:
            flags |= _lib.TRAIN_NORM_LR
        if self.deterministic:
            flags |= _lib.TRAIN_DETERMINISTIC
        if self.update_mode == "atomic":
            flags |= _lib.TRAIN_ATOMIC
        elif self.update_mode == "write_back":
            flags |= _lib.TRAIN_WRITE_BACK
        elif self.update_mode == "write_through":
            flags |= _lib.TRAIN_WRITE_THROUGH
        # GN2V_NEGATIVES=global: the reference-semantic schedule for models that did not choose
        # themselves -- walk-ordered kernels, every negative the endpoint of a uniform random
        # edge of the WHOLE graph (node2vec_skipgram.py:101-102) -- instead of the block path's
        # cell-local draw; reachable from the drop-in classes (which have no such kwarg) through
        # the environment.  "cell" (or unset): the engine decides.
        block_path = self.block_path
        if block_path is None:
            law = os.environ.get("GN2V_NEGATIVES", "").strip().lower()
            if law == "global":
                block_path = False
            elif law not in ("", "cell", "auto"):
                raise ValueError(f"GN2V_NEGATIVES must be 'global' or 'cell', got {law!r}.")
        if block_path is True and self.MODEL_ID == _lib.MODEL_SKIPGRAM:
            flags |= _lib.TRAIN_BLOCK_PATH
        elif block_path is False:
            flags |= _lib.TRAIN_WALK_ORDERED
        return _lib.TrainParams(
            self.MODEL_ID, self.embedding_size, self.padded_size, self.epochs,
            self.number_of_negative_samples, self.window_size, self.learning_rate,
            self.learning_rate_decay, self.clipping_value, flags, self.init_scale(),
            self.min_distance,
        )

    def init_scale(self) -> float:
        """Both tables start uniform in +-1/sqrt(d) ("Initialisation")."""
        return float(1.0 / np.sqrt(self.embedding_size))

    # ------------------------------------------------------------------ the hot call
    def fit_transform_device(self, graph, max_walks_per_epoch: int = 0):
        """Run the fit and leave both tables in HBM: returns (central, contextual) torch tensors
        of shape [N, padded_size] (columns >= embedding_size are zero padding) and the stats."""
        import torch

        csr = _as_csr(graph, self.normalize_by_degree)
        _lib.require_device()
        if not torch.cuda.is_available():
            raise RuntimeError("PyTorch does not see a ROCm device; cannot allocate the tables.")
        dev = torch.device("cuda", self.device)
        dgraph = csr.device_graph(self.device)
        n, ld = csr.get_number_of_nodes(), self.padded_size
        with torch.cuda.device(dev):
            central = torch.empty((n, ld), dtype=torch.float32, device=dev)
            contextual = torch.empty((n, ld), dtype=torch.float32, device=dev)
            # the library allocates its round buffers itself (block path): hand back what
            # PyTorch's caching allocator is only keeping for later
            torch.cuda.empty_cache()
            stream = torch.cuda.current_stream().cuda_stream
            wp, tp, stats = self.walk_params(), self.train_params(), _lib.Stats()
            L = _lib.lib()
            _lib.check(L.gn2v_stats_reset(dgraph.handle, stream))
            start = time.perf_counter()
            _lib.check(L.gn2v_train(dgraph.handle, C.byref(wp), C.byref(tp), self.random_state,
                                    max_walks_per_epoch, central.data_ptr(),
                                    contextual.data_ptr(), C.byref(stats), stream))
            self.last_seconds = time.perf_counter() - start
            # the library keeps the round buffers of a block fit (tens of GB) on the graph handle
            # for the handle's next fit; a model gives them back unless told otherwise
            # (model.keep_buffers = True: repeated fits on one graph skip 1.5-2 s of allocation)
            if not getattr(self, "keep_buffers", False):
                _lib.check(L.gn2v_graph_release_buffers(dgraph.handle))
        self.last_stats = stats.as_dict()
        self.last_plan = ({"world": 1, "parts": stats.block_parts, "slices": stats.block_slices,
                           "stripes": stats.block_stripes,
                           "group_parts": stats.block_group_parts,
                           "round_walks": stats.block_round_walks} if stats.block_parts else None)
        if self.verbose:
            secs = max(self.last_seconds, 1e-9)
            path = (f" (block path, {stats.block_parts} parts x {stats.block_slices} slices, "
                    f"{stats.block_stripes} centre stripes)"
                    if stats.block_parts else "")
            print(
                f"[gn2v] {self.NAME}{path}: {stats.pairs} pairs, {stats.walk_steps} walk steps in "
                f"{secs:.3f}s ({stats.pairs / secs:.3e} pairs/s; train kernels "
                f"{stats.train_ms:.1f} ms, walk kernels {stats.walk_ms:.1f} ms)",
                file=sys.stderr,
            )
        return central, contextual, self.last_stats

    # SkipGram on graphs of >= GN2V_BLOCK_PATH_MIN_NODES nodes in the default update mode is
    # trained through the block path on one GPU as well (gn2v_train decides; contextual rows in
    # XCD-exclusive cells: 1.0 instead of 0.70 of the HBM roofline at 10 M nodes, 22 x the speed
    # of atomics at 2 708 nodes, link quality at or above the walk-ordered schedule's; DESIGN.md
    # section 7)
    BLOCK_PATH_MIN_NODES = _lib.BLOCK_PATH_MIN_NODES

    def fit_transform_blocks(self, graph, comm, round_walks: Optional[int] = None, slices=None,
                             parts=None, overlap: bool = True, max_walks_per_epoch: int = 0,
                             stripes: Optional[int] = None, group_parts: Optional[int] = None,
                             root: Optional[int] = None):
        """SkipGram over several GPUs, one process per GPU (``comm`` = ``distributed.TorchComm``
        under ``torch.distributed.run``): tables partitioned by node id, no row shared between
        GPUs (``distributed.BlockPartitionedTrainer``).  Every rank returns the full
        ``(central, contextual)`` device tensors [N, padded_size] -- or, with ``root``, that rank
        only (the others return ``(None, None)`` and never hold more than their shards).
        ``group_parts``: parts whose pairs are extracted and held at a time (None: from the free
        HBM, ``distributed.round_plan``).  With one rank
        (``LoopbackComm``) this is the Python form of ``gn2v_train_blocks``: ``stripes`` centre
        stripes (None = 1 = none, as there) play the ranks one after the other."""
        import torch

        from . import ops
        from .distributed import BlockPartitionedTrainer, round_plan

        if self.MODEL_ID != _lib.MODEL_SKIPGRAM:
            raise NotImplementedError(
                "Multi-GPU training is available for SkipGram only: CBOW needs all the contexts "
                "of a centre on one GPU."
            )
        csr = _as_csr(graph, self.normalize_by_degree)
        _lib.require_device()
        device = torch.cuda.current_device() if comm.world > 1 else self.device
        dev = torch.device("cuda", device)
        tp = self.train_params()
        L = self.walk_length
        if comm.world > 1:
            stripes = 1
        else:
            stripes = 1 if stripes is None else max(1, int(stripes))
            while stripes > 1 and csr.get_number_of_nodes() // stripes < 2:
                stripes //= 2
        lanes = comm.world if comm.world > 1 else stripes  # ranks, or the stripes that play them
        with torch.cuda.device(dev):
            ops.stats_reset(csr, device)
            trainer = BlockPartitionedTrainer(
                csr, tp, self.embedding_size, self.padded_size, self.random_state,
                self.init_scale(), comm, dev, walk_length=L, window=self.window_size,
                min_dist=self.min_distance, scale_free=self.use_scale_free_distribution,
                slices=slices, parts=parts, stripes=stripes)
            wp = self.walk_params()
            walks_per_epoch = csr.get_number_of_unique_source_nodes() * self.iterations
            if max_walks_per_epoch:
                walks_per_epoch = min(walks_per_epoch, max_walks_per_epoch)
            if round_walks is None or group_parts is None:
                # the full tables a rank returns must fit beside the last rounds
                torch.cuda.empty_cache()
                holds = comm.world > 1 and (root is None or root == comm.rank)
                result = (2 if holds else 0) * trainer.n_nodes * self.padded_size * 4
                cap = 0 if round_walks is None else int(round_walks)
                if round_walks is None and getattr(trainer, "permute", False):
                    # resident cells: 192 rounds -- sets of cell-mates -- over the fit, 16 to 64
                    # per epoch of the graph, none shorter than 2^14 walks (gn2v_train_blocks:
                    # the same rule and the measurement behind it)
                    from .distributed import rounds_per_epoch

                    rounds = rounds_per_epoch(self.epochs)
                    # (several ranks: a launch is one part of one rank: rounds of >= 2^19 walks)
                    shortest = int(os.environ.get("GN2V_ROUND_MIN_WALKS", "")
                                   or (1 << 19 if comm.world > 1 else 1 << 14))
                    epoch_walks = csr.get_number_of_unique_source_nodes() * self.iterations
                    cap = max(shortest, -(-epoch_walks // (rounds * lanes)))
                # (the plan is made for the round that will be trained: its cap)
                auto_walks, auto_group = round_plan(
                    max(0, torch.cuda.mem_get_info(dev)[0] - result), trainer.n_nodes, L,
                    self.window_size, lanes, trainer.parts, trainer.slices,
                    overlap and stripes == 1, cap=cap)
                if comm.world > 1:  # every rank must use the same round size and groups
                    mine = torch.tensor([auto_walks, auto_group], dtype=torch.int64, device=dev)
                    agreed = comm.all_gather(mine).view(-1, 2).min(0).values
                    auto_walks, auto_group = int(agreed[0]), int(agreed[1])
                round_walks = auto_walks if round_walks is None else round_walks
                group_parts = auto_group if group_parts is None else group_parts
            trainer.group_parts = max(1, min(int(group_parts), trainer.parts))
            # round_walks: the walks whose pairs a rank (a stripe) holds at once; a round is
            # `lanes` times that
            round_walks = max(1, min(round_walks, -(-walks_per_epoch // lanes)))
            stride = lanes * round_walks
            trainer.round_capacity = round_walks  # per rank (per stripe pass)
            n_rounds = (walks_per_epoch + stride - 1) // stride
            lr = np.float32(self.learning_rate)
            start = time.perf_counter()
            rounds = []
            for epoch in range(self.epochs):
                for r in range(n_rounds):  # every rank joins every round (collectives inside)
                    first = r * stride
                    share = stride if comm.world == 1 else round_walks  # one rank: all of them
                    mine = first + comm.rank * round_walks
                    n = max(0, min(share, walks_per_epoch - mine))

                    def make(epoch=epoch, mine=mine, n=n, share=share):
                        if comm.world == 1:
                            return ops.walks(csr, wp, self.random_state, epoch, mine, n,
                                             device=device)
                        # ranks with fewer walks left pad with ended (sentinel) walks
                        walks = torch.full((share, L), -1, dtype=torch.int32, device=dev)
                        if n:
                            walks[:n] = ops.walks(csr, wp, self.random_state, epoch, mine, n,
                                                  device=device)
                        return walks

                    rounds.append((make, self.random_state, epoch, float(lr), first))
                lr = np.float32(lr * np.float32(self.learning_rate_decay))
            trainer.run(rounds, overlap=overlap)
            central, contextual = trainer.gather_full(root=root)
            torch.cuda.synchronize(dev)
            self.last_seconds = time.perf_counter() - start
        self.last_stats = ops.stats_read(csr, device)
        self.last_plan = {"world": comm.world, "parts": trainer.parts, "slices": trainer.slices,
                          "stripes": trainer.stripes, "group_parts": trainer.group_parts,
                          "round_walks": round_walks}
        if self.verbose and comm.rank == 0:
            st, secs = self.last_stats, max(self.last_seconds, 1e-9)
            print(
                f"[gn2v] {self.NAME} (block path, {comm.world} GPU(s), {trainer.parts} parts x "
                f"{trainer.slices} slices): {st['pairs']} pairs on this rank in {secs:.3f}s "
                f"({st['pairs'] / secs:.3e} pairs/s; train kernels {st['train_ms']:.1f} ms)",
                file=sys.stderr,
            )
        return central, contextual

    def fit_transform_world(self, graph, comm, round_walks: int = 0, max_walks_per_epoch: int = 0):
        """SkipGram over several GPUs through the C loop (``gn2v_train_world``, include/gn2v.h):
        the schedule of ``fit_transform_blocks`` without Python between the launches -- what a
        non-Python binding of node2vec.py:99 calls on every rank.  ``comm``: one of
        ``distributed``'s communicators (wrapped as a ``gn2v_comm`` of callbacks; a C host hands
        over RCCL calls instead).  Every rank returns the full ``(central, contextual)`` device
        tensors [N, padded_size]."""
        import torch

        from .distributed import CComm

        if self.MODEL_ID != _lib.MODEL_SKIPGRAM:
            raise NotImplementedError("Multi-GPU training is available for SkipGram only.")
        csr = _as_csr(graph, self.normalize_by_degree)
        _lib.require_device()
        device = torch.cuda.current_device() if comm.world > 1 else self.device
        dev = torch.device("cuda", device)
        dgraph = csr.device_graph(device)
        n, ld = csr.get_number_of_nodes(), self.padded_size
        with torch.cuda.device(dev):
            central = torch.empty((n, ld), dtype=torch.float32, device=dev)
            contextual = torch.empty((n, ld), dtype=torch.float32, device=dev)
            torch.cuda.empty_cache()
            stream = torch.cuda.current_stream().cuda_stream
            wp, tp, stats = self.walk_params(), self.train_params(), _lib.Stats()
            c_comm = CComm(comm, dev)
            L = _lib.lib()
            _lib.check(L.gn2v_stats_reset(dgraph.handle, stream))
            start = time.perf_counter()
            rc = L.gn2v_train_world(dgraph.handle, C.byref(wp), C.byref(tp), self.random_state,
                                    max_walks_per_epoch, int(round_walks),
                                    C.byref(c_comm.struct), central.data_ptr(),
                                    contextual.data_ptr(), C.byref(stats), stream)
            if rc and c_comm.error is not None:
                raise c_comm.error
            _lib.check(rc)
            self.last_seconds = time.perf_counter() - start
            if not getattr(self, "keep_buffers", False):
                _lib.check(L.gn2v_graph_release_buffers(dgraph.handle))
        self.last_stats = stats.as_dict()
        self.last_plan = {"world": comm.world, "parts": stats.block_parts,
                          "slices": stats.block_slices, "stripes": 1,
                          "group_parts": stats.block_group_parts,
                          "round_walks": stats.block_round_walks}
        return central, contextual

    def fit_transform(self, graph) -> List[np.ndarray]:
        """``[central, contextual]`` as freshly allocated C-contiguous float32 [N, d] arrays
        (or ``np.memmap``s when the ``*_embedding_path`` arguments are given).

        Multi-GPU is an explicit opt-in: set ``model.comm`` to a ``distributed.TorchComm`` (or
        export ``GN2V_DISTRIBUTED=1`` inside an initialised ``torch.distributed`` job: the default
        process group is then used) and every rank must make the call.  Models without a
        multi-GPU path (CBOW, GloVe) always run on their own ``device``."""
        central = contextual = None
        comm = getattr(self, "comm", None)
        if comm is None and os.environ.get("GN2V_DISTRIBUTED", "0") not in ("", "0"):
            import torch.distributed as dist

            if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
                from .distributed import TorchComm

                comm = TorchComm()
        if comm is not None and comm.world > 1 and self.MODEL_ID == _lib.MODEL_SKIPGRAM:
            # model.gather_root = r: only rank r assembles (and returns) the tables, the other
            # ranks return None -- 1 / world of the result traffic and no full tables beside
            # their shards; default (None): every rank returns them, like the reference's call
            root = getattr(self, "gather_root", None)
            if os.environ.get("GN2V_WORLD_LOOP", "python") == "c":
                # the rounds of every rank driven by gn2v_train_world, the communicator as callbacks
                central, contextual = self.fit_transform_world(graph, comm)
            else:
                central, contextual = self.fit_transform_blocks(graph, comm, root=root)
            if central is None:
                return None
        if central is None:
            central, contextual, _ = self.fit_transform_device(graph)
        return self._download(central, contextual)

    def _download(self, central, contextual) -> List[np.ndarray]:
        """Device tables [N, padded] f32 -> host arrays [N, d] of the model's ``dtype`` (converted
        on the device, so an f16 result crosses PCIe at half the size)."""
        import torch

        d = self.embedding_size
        as_type = getattr(torch, _RESULT_DTYPES[self.dtype])
        out = []
        for tensor, path in ((central, self.central_nodes_embedding_path),
                             (contextual, self.contextual_nodes_embedding_path)):
            host = tensor[:, :d].to(as_type).contiguous().cpu().numpy()
            if path is not None:
                os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
                mm = np.lib.format.open_memmap(path, mode="w+", dtype=host.dtype,
                                               shape=host.shape)
                mm[:] = host
                mm.flush()
                host = mm
            out.append(host)
        return out


class _WalkletsModel:
    """Walklets: one independent table pair per scale s = 1..window_size, trained only on the
    (centre, context) pairs exactly s steps apart in the walks, each of size `embedding_size`
    (the wrapper passes embedding_size // window_size, embedders/ensmallen_embedders/walklets.py:113).
    ``fit_transform`` returns ``[central_1, contextual_1, ..., central_w, contextual_w]``; path
    arguments may contain ``{window_size}``, replaced by the scale (walklets.py:79-90)."""

    BASE = None

    def __init__(self, embedding_size: int = 100, random_state: int = 42, window_size: int = 4,
                 central_nodes_embedding_path: Optional[str] = None,
                 contextual_nodes_embedding_path: Optional[str] = None, **kwargs):
        self.window_size = int(window_size)
        self._paths = (central_nodes_embedding_path, contextual_nodes_embedding_path)
        self._scales = []
        kwargs.setdefault("verbose", False)  # the Walklets wrappers have no verbose argument
        for s in range(1, self.window_size + 1):
            paths = [None if p is None else p.replace("{window_size}", str(s))
                     for p in self._paths]
            self._scales.append(self.BASE(
                embedding_size=embedding_size, random_state=random_state, window_size=s,
                min_distance=s, central_nodes_embedding_path=paths[0],
                contextual_nodes_embedding_path=paths[1], **kwargs))
        self.last_stats = None

    @property
    def random_state(self):
        return self._scales[0].random_state

    @random_state.setter
    def random_state(self, value):
        for m in self._scales:
            m.random_state = int(value)

    @property
    def comm(self):
        return getattr(self._scales[0], "comm", None)

    @comm.setter
    def comm(self, value):
        for m in self._scales:
            m.comm = value

    @property
    def deterministic(self):
        return self._scales[0].deterministic

    @deterministic.setter
    def deterministic(self, value):
        for m in self._scales:
            m.deterministic = bool(value)

    def fit_transform(self, graph) -> List[np.ndarray]:
        out, stats = [], []
        for m in self._scales:
            out.extend(m.fit_transform(graph))
            stats.append(m.last_stats)
        self.last_stats = {
            key: sum(st[key] for st in stats) for key in stats[0]
        } if stats else None
        return out


class SkipGram(_WalkBasedModel):
    MODEL_ID = _lib.MODEL_SKIPGRAM
    NAME = "SkipGram"


class CBOW(_WalkBasedModel):
    MODEL_ID = _lib.MODEL_CBOW
    NAME = "CBOW"


class GloVe(_WalkBasedModel):
    """GloVe on the co-occurrences of the walks (``models.GloVe`` of the reference's table,
    embedders/ensmallen_embedders/node2vec.py:16-26; kwargs node2vec_glove.py:8-30: ``alpha``,
    100 epochs, walks of 512 nodes, one iteration, learning rate 0.05 x 0.9 per epoch).  Semantics
    (published GloVe under these kwargs) are stated in oracle/gn2v_oracle.c; the co-occurrence
    matrix is counted once from ``iterations`` walks per source node, then trained for ``epochs``
    passes over its non-zero entries."""

    MODEL_ID = 2
    NAME = "GloVe"
    SLOTS_PER_BATCH = 1 << 27  # co-occurrence slots generated and reduced at a time (2 x 1 GiB)

    def update_flags(self) -> int:
        if self.deterministic:
            return _lib.TRAIN_DETERMINISTIC
        return {"auto": 0, "atomic": _lib.TRAIN_ATOMIC, "write_back": _lib.TRAIN_WRITE_BACK,
                "write_through": _lib.TRAIN_WRITE_THROUGH}[self.update_mode]

    def cooccurrence_device(self, csr):
        """Reduced co-occurrence counts of this model's walks: (keys int64, counts int64)."""
        from . import cooccurrence, ops

        wp = self.walk_params()
        total = csr.get_number_of_unique_source_nodes() * self.iterations
        per_walk = self.walk_length * 2 * self.window_size
        batch = max(1, self.SLOTS_PER_BATCH // per_walk)
        acc = cooccurrence.Accumulator()
        for first in range(0, total, batch):
            walks = ops.walks(csr, wp, self.random_state, 0, first, min(batch, total - first),
                              device=self.device)
            keys, weights = ops.cooc_slots(walks, self.window_size, self.min_distance)
            acc.add(cooccurrence.reduce_slots(keys, weights))
            del walks, keys, weights
        return acc.result()

    def fit_transform_device(self, graph, max_walks_per_epoch: int = 0):
        import torch

        from . import cooccurrence, ops

        csr = _as_csr(graph, self.normalize_by_degree)
        _lib.require_device()
        if not torch.cuda.is_available():
            raise RuntimeError("PyTorch does not see a ROCm device; cannot allocate the tables.")
        dev = torch.device("cuda", self.device)
        n, d, ld = csr.get_number_of_nodes(), self.embedding_size, self.padded_size
        with torch.cuda.device(dev):
            ops.stats_reset(csr, self.device)
            start = time.perf_counter()
            keys, counts = self.cooccurrence_device(csr)
            n_entries = int(keys.numel())  # record slots minus padding
            rows, cols, logx, fx = cooccurrence.entries(keys, counts, self.random_state, self.alpha)
            del keys, counts
            central = ops.init_table(n, d, self.random_state, 0, self.init_scale(), self.device, ld)
            contextual = ops.init_table(n, d, self.random_state, 1, self.init_scale(), self.device, ld)
            bias_c = torch.zeros(n, dtype=torch.float32, device=dev)
            bias_x = torch.zeros(n, dtype=torch.float32, device=dev)
            lr, flags = np.float32(self.learning_rate), self.update_flags()
            for _ in range(self.epochs):
                ops.glove_step(csr, rows, cols, logx, fx, central, contextual, bias_c, bias_x, d,
                               float(lr), flags)
                lr = np.float32(lr * np.float32(self.learning_rate_decay))
            torch.cuda.synchronize(dev)
            self.last_seconds = time.perf_counter() - start
        stats = ops.stats_read(csr, self.device)
        stats["entries"] = n_entries
        stats["pairs"] = stats["entries"] * self.epochs  # entry updates
        self.last_stats = stats
        if self.verbose:
            print(f"[gn2v] GloVe: {stats['entries']} co-occurrence entries x {self.epochs} epochs in "
                  f"{self.last_seconds:.3f}s (train kernels {stats['train_ms']:.1f} ms)",
                  file=sys.stderr)
        return central, contextual, stats

    def fit_transform_blocks(self, graph, comm, round_walks: int = 1 << 18):
        raise NotImplementedError("Multi-GPU training is available for SkipGram only.")


class WalkletsSkipGram(_WalkletsModel):
    BASE = SkipGram


class WalkletsCBOW(_WalkletsModel):
    BASE = CBOW


class WalkletsGloVe(_WalkletsModel):
    BASE = GloVe

"""CSR graph container that stands in for ``ensmallen.Graph`` on the Node2Vec hot path.

It exposes exactly the getters the reference touches on this path
(embiggen/utils/abstract_models/abstract_embedding_model.py:114-180,:236;
embedders/ensmallen_embedders/node2vec.py:105; embedders/graph_embedding_pipeline.py:91-92) with
the same names, and stores the graph in the CSR convention the reference exports to sibling
back-ends (embedders/pecanpy_embedders/node2vec.py:139-163): ``row_ptr`` = zero-prefixed
cumulative node degrees, ``col_idx`` = directed destination ids, optional positive edge weights.

Layout: ``row_ptr`` u64[N+1], ``col_idx`` u32[E_directed] with neighbours ascending and unique
per row, ``cumw`` f32[E] per-row inclusive prefix sums of the weights (weighted graphs only),
``sources`` u32 = nodes with out-degree > 0.  Arrays live either on the host (numpy) or, for the
large synthetic graphs, directly in HBM (torch tensors, never copied back).
"""
import ctypes as C
import os
import threading
from typing import List, Optional

import numpy as np

from . import _lib

_DATA = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data")
_HANDLE_LOCK = threading.Lock()


class DeviceGraph:
    """Owner of a ``gn2v_graph*`` handle (and of the tensors it borrows)."""

    def __init__(self, handle, device: int, keep_alive=()):
        self.handle = handle
        self.device = device
        self._keep_alive = keep_alive

    def close(self):
        if self.handle is not None:
            _lib.lib().gn2v_graph_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class CSRGraph:
    def __init__(self, row_ptr, col_idx, weights=None, node_names: Optional[List[str]] = None,
                 name: str = "Graph", directed: bool = False, _device_tensors=None):
        self._name = name
        self._directed = directed
        self._node_names = node_names
        self._device_tensors = _device_tensors  # dict of torch cuda tensors or None
        self._handles = {}
        if _device_tensors is not None:
            self._row_ptr = self._col_idx = self._weights = self._cumw = None
            self._n_nodes = int(_device_tensors["row_ptr"].numel()) - 1
            self._n_edges = int(_device_tensors["col_idx"].numel())
            self._n_sources = int(_device_tensors["n_sources"])
            return
        self._row_ptr = np.ascontiguousarray(row_ptr, dtype=np.uint64)
        self._col_idx = np.ascontiguousarray(col_idx, dtype=np.uint32)
        self._n_nodes = len(self._row_ptr) - 1
        self._n_edges = len(self._col_idx)
        if self._n_nodes >= 0xFFFFFFFF:
            raise ValueError("Node ids must fit in 32 bits.")
        if int(self._row_ptr[-1]) != self._n_edges or int(self._row_ptr[0]) != 0:
            raise ValueError("row_ptr is not a valid CSR offsets array for col_idx.")
        if self._n_edges and int(self._col_idx.max()) >= self._n_nodes:
            raise ValueError("col_idx contains a node id outside the graph.")
        self._weights = None
        self._cumw = None
        if weights is not None:
            self._weights = np.ascontiguousarray(weights, dtype=np.float32)
            if len(self._weights) != self._n_edges:
                raise ValueError("weights must have one entry per directed edge.")
            cum = np.cumsum(self._weights.astype(np.float64))
            starts = self._row_ptr[:-1].astype(np.int64)
            base = np.concatenate([[0.0], cum])[starts]
            deg = np.diff(self._row_ptr.astype(np.int64))
            self._cumw = (cum - np.repeat(base, deg)).astype(np.float32)
        deg = np.diff(self._row_ptr.astype(np.int64))
        self._sources = np.flatnonzero(deg > 0).astype(np.uint32)
        self._n_sources = len(self._sources)
        if node_names is not None and len(node_names) != self._n_nodes:
            raise ValueError("node_names must have one entry per node.")

    # ------------------------------------------------------------------ constructors
    @classmethod
    def from_edge_list(cls, sources, destinations, weights=None, number_of_nodes=None,
                       directed: bool = False, node_names=None, name: str = "Graph"):
        """Build from an edge list; undirected edges are stored in both directions, neighbour
        lists are sorted and duplicate edges collapsed (weights of duplicates are summed)."""
        src = np.asarray(sources, dtype=np.int64).ravel()
        dst = np.asarray(destinations, dtype=np.int64).ravel()
        if src.shape != dst.shape:
            raise ValueError("sources and destinations must have the same length.")
        if number_of_nodes is None:
            number_of_nodes = (
                len(node_names) if node_names is not None
                else (int(max(src.max(), dst.max())) + 1 if len(src) else 0)
            )
        n = int(number_of_nodes)
        if len(src) and (src.min() < 0 or dst.min() < 0 or max(src.max(), dst.max()) >= n):
            raise ValueError("Edge list contains node ids outside [0, number_of_nodes).")
        w = None if weights is None else np.asarray(weights, dtype=np.float64).ravel()
        if not directed:
            loops = src == dst
            s2 = np.concatenate([src, dst[~loops]])
            d2 = np.concatenate([dst, src[~loops]])
            if w is not None:
                w = np.concatenate([w, w[~loops]])
            src, dst = s2, d2
        key = src * n + dst
        if w is None:
            key = np.unique(key)
        else:
            key, inv = np.unique(key, return_inverse=True)
            w = np.bincount(inv, weights=w, minlength=len(key))
        rows = key // max(n, 1)
        cols = key - rows * n
        row_ptr = np.zeros(n + 1, dtype=np.uint64)
        np.cumsum(np.bincount(rows, minlength=n), out=row_ptr[1:])
        return cls(row_ptr, cols.astype(np.uint32), w, node_names, name, directed)

    @classmethod
    def from_csr(cls, row_ptr, col_idx, weights=None, node_names=None, name: str = "Graph",
                 directed: bool = False):
        """Wrap existing CSR arrays (neighbours must be ascending and unique per row)."""
        return cls(row_ptr, col_idx, weights, node_names, name, directed)

    @classmethod
    def from_networkx(cls, graph, weight: Optional[str] = None, name: Optional[str] = None):
        nodes = list(graph.nodes())
        index = {u: i for i, u in enumerate(nodes)}
        src, dst, w = [], [], []
        for u, v, data in graph.edges(data=True):
            src.append(index[u])
            dst.append(index[v])
            if weight is not None:
                w.append(float(data.get(weight, 1.0)))
        return cls.from_edge_list(
            src, dst, w if weight is not None else None, number_of_nodes=len(nodes),
            directed=graph.is_directed(), node_names=[str(u) for u in nodes],
            name=name or (graph.name if getattr(graph, "name", "") else "Graph"),
        )

    @classmethod
    def from_ensmallen(cls, graph):
        """Adapter for users who do have the ensmallen wheel: uses the CSR getters the reference
        itself uses at embedders/pecanpy_embedders/node2vec.py:144-163."""
        n = graph.get_number_of_nodes()
        row_ptr = np.zeros(n + 1, dtype=np.uint64)
        row_ptr[1:] = graph.get_cumulative_node_degrees().astype(np.uint64)
        col_idx = graph.get_directed_destination_node_ids().astype(np.uint32)
        weights = graph.get_directed_edge_weights() if graph.has_edge_weights() else None
        return cls(row_ptr, col_idx, weights, list(graph.get_node_names()), graph.get_name(),
                   graph.is_directed())

    # ------------------------------------------------------------------ ensmallen.Graph getters
    def get_name(self) -> str:
        return self._name

    def has_nodes(self) -> bool:
        return self._n_nodes > 0

    def has_edges(self) -> bool:
        return self._n_edges > 0

    def is_directed(self) -> bool:
        return self._directed

    def get_number_of_nodes(self) -> int:
        return self._n_nodes

    def get_number_of_directed_edges(self) -> int:
        return self._n_edges

    def get_number_of_unique_source_nodes(self) -> int:
        """`sample_number` of Node2VecSequence (node2vec_sequence.py:88)."""
        return self._n_sources

    def get_node_names(self) -> List[str]:
        if self._node_names is None:
            self._node_names = [str(i) for i in range(self._n_nodes)]
        return self._node_names

    def has_node_types(self) -> bool:
        return False

    def get_number_of_node_types(self) -> int:
        return 0

    def has_edge_types(self) -> bool:
        return False

    def has_edge_weights(self) -> bool:
        return self._weights is not None or (
            self._device_tensors is not None and "cumw" in self._device_tensors)

    def has_negative_edge_weights(self) -> bool:
        return self._weights is not None and bool((self._weights <= 0).any())

    def get_node_degrees(self) -> np.ndarray:
        if self._device_tensors is not None:
            rp = self._device_tensors["row_ptr"]
            return (rp[1:] - rp[:-1]).cpu().numpy()
        return np.diff(self._row_ptr.astype(np.int64))

    def get_cumulative_node_degrees(self) -> np.ndarray:
        return self.row_ptr[1:]

    def get_directed_destination_node_ids(self) -> np.ndarray:
        return self.col_idx

    def get_directed_edge_weights(self) -> np.ndarray:
        if self._weights is None:
            raise ValueError("The graph does not have edge weights.")
        return self._weights

    def get_number_of_disconnected_nodes(self) -> int:
        """Nodes with neither outgoing nor incoming edges."""
        if self._device_tensors is not None:
            return self._n_nodes - self._n_sources
        deg = self.get_node_degrees()
        if not self._directed:
            return int((deg == 0).sum())
        indeg = np.bincount(self._col_idx, minlength=self._n_nodes)
        return int(((deg == 0) & (indeg == 0)).sum())

    def has_disconnected_nodes(self) -> bool:
        return self.get_number_of_disconnected_nodes() > 0

    def has_nodes_sorted_by_decreasing_outbound_node_degree(self) -> bool:
        deg = self.get_node_degrees()
        return bool((deg[:-1] >= deg[1:]).all())

    def sort_by_decreasing_outbound_node_degree(self) -> "CSRGraph":
        """New graph with node ids re-labelled by decreasing out-degree (stable)."""
        if self._device_tensors is not None:
            raise NotImplementedError("Sorting a device-resident graph is not supported.")
        deg = self.get_node_degrees()
        order = np.argsort(-deg, kind="stable")  # new id -> old id
        new_of_old = np.empty_like(order)
        new_of_old[order] = np.arange(self._n_nodes)
        src_old = np.repeat(np.arange(self._n_nodes), deg)
        names = self._node_names
        g = CSRGraph.from_edge_list(
            new_of_old[src_old], new_of_old[self._col_idx.astype(np.int64)], self._weights,
            number_of_nodes=self._n_nodes, directed=True,
            node_names=None if names is None else [names[i] for i in order], name=self._name,
        )
        g._directed = self._directed
        return g

    def with_degree_normalized_weights(self) -> "CSRGraph":
        """Same graph with every edge weight divided by the degree of its destination node: the
        walk sampler then realises ``normalize_by_degree`` ("normalize the random walk by the node
        degree of the destination node", node2vec_skipgram.py:94-96) through its ordinary
        weight-proportional candidate draw.  The derived graph is cached on this object."""
        cached = getattr(self, "_degree_normalized", None)
        if cached is not None:
            return cached
        if self._device_tensors is not None:
            import torch

            t = self._device_tensors
            rp, col = t["row_ptr"], t["col_idx"].to(torch.int64) & 0xFFFFFFFF
            deg = (rp[1:] - rp[:-1])
            w = 1.0 / deg[col].to(torch.float64)
            cum = torch.cumsum(w, 0)
            base = torch.cat([torch.zeros(1, dtype=cum.dtype, device=cum.device), cum])[rp[:-1]]
            cumw = (cum - torch.repeat_interleave(base, deg)).to(torch.float32)
            g = CSRGraph(None, None, name=self._name, directed=self._directed,
                         _device_tensors=dict(t, cumw=cumw))
            g._node_names = self._node_names
        else:
            deg = np.diff(self._row_ptr.astype(np.int64)).astype(np.float64)
            base = np.ones(self._n_edges) if self._weights is None else self._weights.astype(np.float64)
            w = base / np.maximum(deg[self._col_idx.astype(np.int64)], 1.0)
            g = CSRGraph(self._row_ptr, self._col_idx, w, self._node_names, self._name,
                         self._directed)
        self._degree_normalized = g
        return g

    # ------------------------------------------------------------------ raw arrays
    @property
    def row_ptr(self) -> np.ndarray:
        if self._row_ptr is None:
            self._row_ptr = self._device_tensors["row_ptr"].cpu().numpy().astype(np.uint64)
        return self._row_ptr

    @property
    def col_idx(self) -> np.ndarray:
        if self._col_idx is None:
            self._col_idx = self._device_tensors["col_idx"].cpu().numpy().view(np.uint32)
        return self._col_idx

    @property
    def cumw(self) -> Optional[np.ndarray]:
        if self._cumw is None and self._device_tensors is not None and "cumw" in self._device_tensors:
            self._cumw = self._device_tensors["cumw"].cpu().numpy()
        return self._cumw

    @property
    def sources(self) -> Optional[np.ndarray]:
        """u32 ids of the nodes walks start from, or None when every node is a source."""
        if self._n_sources == self._n_nodes:
            return None
        if self._device_tensors is not None:
            return self._device_tensors["sources"].cpu().numpy().view(np.uint32)
        return self._sources

    # ------------------------------------------------------------------ device side
    def device_graph(self, device: int = 0) -> DeviceGraph:
        """Create (once per device) the engine-side handle; uploads host arrays to HBM."""
        with _HANDLE_LOCK:
            return self._device_graph_locked(device)

    def _device_graph_locked(self, device: int) -> DeviceGraph:
        if device in self._handles:
            return self._handles[device]
        L = _lib.lib()
        _lib.require_device()
        handle = C.c_void_p()
        if self._device_tensors is not None:
            t = self._device_tensors
            if t["row_ptr"].device.index != device:
                raise ValueError("This graph lives on another device.")
            src = t.get("sources")
            cumw = t.get("cumw")
            _lib.check(L.gn2v_graph_create(
                t["row_ptr"].data_ptr(), t["col_idx"].data_ptr(),
                None if cumw is None else cumw.data_ptr(),
                None if src is None else src.data_ptr(), self._n_nodes, self._n_edges,
                self._n_sources, _lib.GRAPH_DEVICE_PTRS, device, C.byref(handle)))
            dg = DeviceGraph(handle, device, keep_alive=(t,))
        else:
            srcs = self.sources
            _lib.check(L.gn2v_graph_create(
                self._row_ptr.ctypes.data, self._col_idx.ctypes.data,
                None if self._cumw is None else self._cumw.ctypes.data,
                None if srcs is None else srcs.ctypes.data, self._n_nodes, self._n_edges,
                self._n_sources, 0, device, C.byref(handle)))
            dg = DeviceGraph(handle, device)
        self._handles[device] = dg
        return dg


def karate_club() -> CSRGraph:
    """Zachary's Karate Club (34 nodes / 78 edges), BASELINE config 1."""
    edges = np.loadtxt(os.path.join(_DATA, "karate.edges"), dtype=np.int64)
    return CSRGraph.from_edge_list(edges[:, 0], edges[:, 1], number_of_nodes=34, name="KarateClub")


def barabasi_albert(number_of_nodes: int, m: int, seed: int = 42, device: int = 0,
                    name: Optional[str] = None) -> CSRGraph:
    """Seeded Barabasi-Albert graph generated and kept in HBM (synthetic benchmark graphs of
    BASELINE.md section 3).  Edges come from the engine's ``gn2v_ba_edges`` kernel; the CSR is
    assembled with torch sort/unique on the device (plumbing only)."""
    import torch

    _lib.require_device()
    L = _lib.lib()
    n = int(number_of_nodes)
    n_e = (n - 1) * m
    dev = torch.device("cuda", device)
    with torch.cuda.device(dev):
        src = torch.empty(n_e, dtype=torch.int32, device=dev)
        dst = torch.empty(n_e, dtype=torch.int32, device=dev)
        stream = torch.cuda.current_stream().cuda_stream
        _lib.check(L.gn2v_ba_edges(n, m, seed, src.data_ptr(), dst.data_ptr(), stream))
        s64, d64 = src.long(), dst.long()
        del src, dst
        key = torch.cat([s64 * n + d64, d64 * n + s64])
        del s64, d64
        key = torch.unique(key)  # sorted, duplicates (multi-edges) collapsed
        rows = torch.div(key, n, rounding_mode="floor")
        cols = (key - rows * n).to(torch.int32)
        del key
        counts = torch.bincount(rows, minlength=n)
        del rows
        row_ptr = torch.zeros(n + 1, dtype=torch.int64, device=dev)
        torch.cumsum(counts, 0, out=row_ptr[1:])
        n_sources = int((counts > 0).sum().item())
        tensors = {"row_ptr": row_ptr, "col_idx": cols, "n_sources": n_sources}
        if n_sources != n:
            tensors["sources"] = torch.nonzero(counts > 0).flatten().to(torch.int32)
        torch.cuda.synchronize(dev)
    return CSRGraph(None, None, name=name or f"BA_{n}_{m}_{seed}", _device_tensors=tensors)

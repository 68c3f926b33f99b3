"""CSR graph container that stands in for ``ensmallen.Graph`` on the Node2Vec hot path.

It exposes exactly the getters the reference touches on this path
(embiggen/utils/abstract_models/abstract_embedding_model.py:114-180,:236;
embedders/ensmallen_embedders/node2vec.py:105; embedders/graph_embedding_pipeline.py:91-92) with
the same names, and stores the graph in the CSR convention the reference exports to sibling
back-ends (embedders/pecanpy_embedders/node2vec.py:139-163): ``row_ptr`` = zero-prefixed
cumulative node degrees, ``col_idx`` = directed destination ids, optional positive edge weights.

Layout: ``row_ptr`` u64[N+1], ``col_idx`` u32[E_directed] with neighbours ascending and unique
per row (typed multigraphs list a neighbour once per edge type), ``cumw`` f32[E] per-row inclusive
prefix sums of the weights (weighted graphs only), ``sources`` u32 = nodes with out-degree > 0,
optional ``node_type_ids`` u32[N] / ``edge_type_ids`` u32[E] (equal id = same type, 0xFFFFFFFF =
unknown; they only feed ``change_node_type_weight`` / ``change_edge_type_weight`` of the walks).
Arrays live either on the host (numpy) or, for the large synthetic graphs, directly in HBM (torch
tensors, never copied back).
"""
import ctypes as C
import os
import threading
from typing import List, Optional

import numpy as np

from . import _lib

_DATA = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data")
_HANDLE_LOCK = threading.Lock()
UNKNOWN_TYPE = 0xFFFFFFFF


def _canonical_type_ids(labels, count: int, what: str):
    """Labels (None = unknown, a hashable, or a list / tuple / set of hashables for multi-label
    nodes) -> (u32 ids, vocabulary, lists).  ``ids``: one id per distinct label *set* (walks only
    compare types for equality, so a label set is one type), ordered by first appearance, unknown
    = 0xFFFFFFFF.  ``vocabulary``: the individual labels by first appearance.  ``lists``: per item
    None or the u32 array of its individual label ids (what ensmallen's get_node_type_ids
    returns)."""
    if len(labels) != count:
        raise ValueError(f"{what} must have one entry per {what.split('_')[0]}.")
    ids = np.empty(count, dtype=np.uint32)
    sets, vocabulary, index, lists = {}, [], {}, []
    for i, label in enumerate(labels):
        if isinstance(label, np.ndarray):
            label = label.tolist()
        if isinstance(label, (list, tuple, set, frozenset)):
            members = list(dict.fromkeys(label))  # ordered, unique
        elif label is None:
            members = []
        else:
            members = [label.item() if isinstance(label, np.generic) else label]
        if not members:
            ids[i] = UNKNOWN_TYPE
            lists.append(None)
            continue
        for m in members:
            if m not in index:
                index[m] = len(vocabulary)
                vocabulary.append(m)
        own = np.array(sorted(index[m] for m in members), dtype=np.uint32)
        lists.append(own)
        ids[i] = sets.setdefault(tuple(own.tolist()), len(sets))
    return ids, vocabulary, lists


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
                 name: str = "Graph", directed: bool = False, _device_tensors=None,
                 node_type_ids=None, edge_type_ids=None, node_type_names=None,
                 edge_type_names=None, node_type_lists=None):
        self._name = name
        self._directed = directed
        self._node_names = node_names
        self._named = node_names is not None  # False: names are the default str(id)
        self._device_tensors = _device_tensors  # dict of torch cuda tensors or None
        self._handles = {}
        self._node_type_ids = self._edge_type_ids = None
        # vocabularies (individual labels) and, for multi-label nodes, the per-node label ids
        self._node_type_names, self._edge_type_names = node_type_names, edge_type_names
        self._node_type_lists = node_type_lists
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
        if node_type_ids is not None:
            self._node_type_ids = np.ascontiguousarray(node_type_ids, dtype=np.uint32)
            if len(self._node_type_ids) != self._n_nodes:
                raise ValueError("node_type_ids must have one entry per node.")
        if edge_type_ids is not None:
            self._edge_type_ids = np.ascontiguousarray(edge_type_ids, dtype=np.uint32)
            if len(self._edge_type_ids) != self._n_edges:
                raise ValueError("edge_type_ids must have one entry per directed edge.")

    # ------------------------------------------------------------------ constructors
    @staticmethod
    def _assemble(src, dst, w, etype, n: int, directed: bool):
        """Edge list -> (row_ptr, cols, weights, edge type ids): undirected edges are stored in
        both directions, neighbour lists sorted, duplicate (src, dst[, type]) edges collapsed
        (weights of duplicates summed)."""
        if not directed:
            loops = src == dst
            src, dst = np.concatenate([src, dst[~loops]]), np.concatenate([dst, src[~loops]])
            if w is not None:
                w = np.concatenate([w, w[~loops]])
            if etype is not None:
                etype = np.concatenate([etype, etype[~loops]])
        if etype is None:
            key = src * n + dst
        else:
            order = np.lexsort((etype, dst, src))
            src, dst, etype = src[order], dst[order], etype[order]
            if w is not None:
                w = w[order]
            new = np.ones(len(src), dtype=bool)
            new[1:] = (src[1:] != src[:-1]) | (dst[1:] != dst[:-1]) | (etype[1:] != etype[:-1])
            key = np.cumsum(new) - 1  # already sorted: unique() keeps this order
        if w is None:
            key, first = np.unique(key, return_index=True)
        else:
            key, first, inv = np.unique(key, return_index=True, return_inverse=True)
            w = np.bincount(inv, weights=w, minlength=len(key))
        if etype is None:
            rows = key // max(n, 1)
            cols = key - rows * n
        else:
            rows, cols, etype = src[first], dst[first], etype[first].astype(np.uint32)
        row_ptr = np.zeros(n + 1, dtype=np.uint64)
        np.cumsum(np.bincount(rows, minlength=n), out=row_ptr[1:])
        return row_ptr, cols.astype(np.uint32), w, etype

    @classmethod
    def from_edge_list(cls, sources, destinations, weights=None, number_of_nodes=None,
                       directed: bool = False, node_names=None, name: str = "Graph",
                       node_types=None, edge_types=None):
        """Build from an edge list; undirected edges are stored in both directions, neighbour
        lists are sorted and duplicate edges collapsed (weights of duplicates are summed).
        ``node_types``: one label per node (None = unknown; a list of labels for multi-label
        nodes).  ``edge_types``: one label per listed edge (None = unknown); parallel edges of
        different types are kept (multigraph), as in ensmallen."""
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
        nt_ids = nt_names = nt_lists = et = et_names = None
        if node_types is not None:
            nt_ids, nt_names, nt_lists = _canonical_type_ids(list(node_types), n, "node_types")
        if edge_types is not None:
            labels = list(edge_types)
            if any(isinstance(t, (list, tuple, set, frozenset, np.ndarray)) for t in labels):
                raise ValueError("An edge has exactly one type (or None).")
            et, et_names, _ = _canonical_type_ids(labels, len(src), "edge_types")
            et = et.astype(np.int64)
        row_ptr, cols, w, et = cls._assemble(src, dst, w, et, n, directed)
        return cls(row_ptr, cols, w, node_names, name, directed, node_type_ids=nt_ids,
                   edge_type_ids=et, node_type_names=nt_names, edge_type_names=et_names,
                   node_type_lists=nt_lists)

    @classmethod
    def from_csr(cls, row_ptr, col_idx, weights=None, node_names=None, name: str = "Graph",
                 directed: bool = False, node_type_ids=None, edge_type_ids=None):
        """Wrap existing CSR arrays (neighbours must be ascending per row, and unique unless
        parallel edges of different types are meant).  ``directed=False`` declares that every edge
        is present in both directions; the walk sampler relies on it."""
        return cls(row_ptr, col_idx, weights, node_names, name, directed,
                   node_type_ids=node_type_ids, edge_type_ids=edge_type_ids)

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
        nt_ids = nt_names = nt_lists = et_ids = None
        if getattr(graph, "has_node_types", lambda: False)():
            # one entry per node: None or the array of its node type ids
            nt_ids, nt_names, nt_lists = _canonical_type_ids(list(graph.get_node_type_ids()), n,
                                                             "node_types")
        if getattr(graph, "has_edge_types", lambda: False)():
            raw = graph.get_directed_edge_type_ids()  # one entry per directed edge, None = unknown
            et_ids = np.array([UNKNOWN_TYPE if t is None else int(t) for t in raw], dtype=np.uint32)
        return cls(row_ptr, col_idx, weights, list(graph.get_node_names()), graph.get_name(),
                   graph.is_directed(), node_type_ids=nt_ids, edge_type_ids=et_ids,
                   node_type_names=nt_names, node_type_lists=nt_lists)

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
        return self.node_type_ids is not None

    def get_number_of_node_types(self) -> int:
        """Distinct known node types (a multi-label set counts as one type here)."""
        return self._count_types(self.node_type_ids)

    def has_edge_types(self) -> bool:
        return self.edge_type_ids is not None

    def get_number_of_edge_types(self) -> int:
        return self._count_types(self.edge_type_ids)

    @staticmethod
    def _count_types(ids) -> int:
        if ids is None:
            return 0
        known = ids[ids != UNKNOWN_TYPE]
        return int(len(np.unique(known)))

    def is_multigraph(self) -> bool:
        """True when some row lists the same neighbour more than once (parallel typed edges)."""
        col, rp = self.col_idx, self.row_ptr.astype(np.int64)
        if len(col) < 2:
            return False
        same = col[1:] == col[:-1]
        same[rp[1:-1][(rp[1:-1] > 0) & (rp[1:-1] < len(col))] - 1] = False  # row boundaries
        return bool(same.any())

    # --- typed-graph getters under ensmallen's names (used by embedding_transformers/
    #     graph_transformer.py:196-243 and node_transformer.py:186-190)
    def get_unique_node_type_names(self) -> List[str]:
        if self._node_type_names is None:
            n = 0 if self.node_type_ids is None else self._count_all(self.node_type_ids)
            self._node_type_names = [str(i) for i in range(n)]
        return self._node_type_names

    def get_unique_edge_type_names(self) -> List[str]:
        if self._edge_type_names is None:
            n = 0 if self.edge_type_ids is None else self._count_all(self.edge_type_ids)
            self._edge_type_names = [str(i) for i in range(n)]
        return self._edge_type_names

    @staticmethod
    def _count_all(ids) -> int:
        known = ids[ids != UNKNOWN_TYPE]
        return int(known.max()) + 1 if len(known) else 0

    def get_node_type_ids(self):
        """Per node: None (unknown) or the u32 array of its node type ids."""
        if not self.has_node_types():
            raise ValueError("The graph does not have node types.")
        if self._node_type_lists is None:
            self._node_type_lists = [None if t == UNKNOWN_TYPE else np.array([t], dtype=np.uint32)
                                     for t in self.node_type_ids]
        return self._node_type_lists

    def get_node_type_ids_from_node_id(self, node_id: int):
        return self.get_node_type_ids()[int(node_id)]

    def get_node_type_names(self):
        """Per node: None or the list of its node type names."""
        vocabulary = self.get_unique_node_type_names()
        return [None if ids is None else [vocabulary[i] for i in ids]
                for ids in self.get_node_type_ids()]

    def get_node_id_from_node_name(self, node_name: str) -> int:
        index = getattr(self, "_node_index", None)
        if index is None:
            index = self._node_index = {n: i for i, n in enumerate(self.get_node_names())}
        if node_name not in index:
            raise ValueError(f"The node name {node_name!r} does not exist in the graph.")
        return index[node_name]

    def get_node_type_names_from_node_name(self, node_name: str):
        ids = self.get_node_type_ids_from_node_id(self.get_node_id_from_node_name(node_name))
        vocabulary = self.get_unique_node_type_names()
        return None if ids is None else [vocabulary[i] for i in ids]

    def get_directed_source_node_ids(self) -> np.ndarray:
        deg = np.diff(self.row_ptr.astype(np.int64))
        return np.repeat(np.arange(self._n_nodes, dtype=np.uint32), deg)

    def _upper_triangular(self) -> np.ndarray:
        return self.get_directed_source_node_ids() <= self.col_idx

    def get_source_node_ids(self, directed: bool = True) -> np.ndarray:
        src = self.get_directed_source_node_ids()
        return src if directed or self._directed else src[self._upper_triangular()]

    def get_destination_node_ids(self, directed: bool = True) -> np.ndarray:
        return self.col_idx if directed or self._directed else self.col_idx[self._upper_triangular()]

    def get_directed_edge_node_names(self):
        names = self.get_node_names()
        return [(names[s], names[d]) for s, d in zip(self.get_directed_source_node_ids(),
                                                     self.col_idx)]

    def _edge_types_or_raise(self) -> np.ndarray:
        if not self.has_edge_types():
            raise ValueError("The graph does not have edge types.")
        return self.edge_type_ids

    def get_directed_edge_type_ids(self):
        return [None if t == UNKNOWN_TYPE else int(t) for t in self._edge_types_or_raise()]

    def get_imputed_directed_edge_type_ids(self, imputation_edge_type_id: int = 0) -> np.ndarray:
        ids = self._edge_types_or_raise()
        return np.where(ids == UNKNOWN_TYPE, np.uint32(imputation_edge_type_id), ids)

    def get_imputed_upper_triangular_edge_type_ids(self, imputation_edge_type_id: int = 0):
        return self.get_imputed_directed_edge_type_ids(imputation_edge_type_id)[
            self._upper_triangular()]

    def get_directed_edge_type_names(self):
        vocabulary = self.get_unique_edge_type_names()
        return [None if t == UNKNOWN_TYPE else vocabulary[t] for t in self._edge_types_or_raise()]

    def get_upper_triangular_edge_type_names(self):
        keep = self._upper_triangular()
        return [t for t, k in zip(self.get_directed_edge_type_names(), keep) if k]

    def has_unknown_edge_types(self) -> bool:
        return self.has_edge_types() and bool((self.edge_type_ids == UNKNOWN_TYPE).any())

    def must_not_contain_unknown_edge_types(self):
        if self.has_unknown_edge_types():
            raise ValueError("The graph contains edges with unknown edge type.")

    def must_not_be_multigraph(self):
        if self.is_multigraph():
            raise ValueError("The graph is a multigraph.")

    # --- label getters of the prediction transformers (ensmallen's names as used by
    #     embedding_transformers/edge_prediction_transformer.py:118-144,
    #     edge_label_prediction_transformer.py:125-229, node_label_prediction_transformer.py:84-150)
    def has_compatible_node_vocabularies(self, other: "CSRGraph") -> bool:
        """Same nodes under the same ids, and node types present in both or in neither."""
        return (self._n_nodes == other.get_number_of_nodes()
                and self.has_node_types() == other.has_node_types()
                and (not (self._named or getattr(other, "_named", True))
                     or self.get_node_names() == other.get_node_names()))

    def has_unknown_node_types(self) -> bool:
        return self.has_node_types() and bool((self.node_type_ids == UNKNOWN_TYPE).any())

    def has_known_node_types(self) -> bool:
        return self.has_node_types() and bool((self.node_type_ids != UNKNOWN_TYPE).any())

    def get_nodes_with_known_node_types_mask(self) -> np.ndarray:
        if not self.has_node_types():
            raise ValueError("The graph does not have node types.")
        return self.node_type_ids != UNKNOWN_TYPE

    def has_multilabel_node_types(self) -> bool:
        return self.has_node_types() and any(
            ids is not None and len(ids) > 1 for ids in self.get_node_type_ids())

    def _node_label_counts(self) -> np.ndarray:
        """Nodes per individual label (a multi-label node counts once for each of its labels)."""
        labels = [ids for ids in self.get_node_type_ids() if ids is not None]
        if not labels:
            return np.zeros(0, dtype=np.int64)
        return np.bincount(np.concatenate(labels).astype(np.int64))

    def has_homogeneous_node_types(self) -> bool:
        return int((self._node_label_counts() > 0).sum()) == 1

    def has_singleton_node_types(self) -> bool:
        return bool((self._node_label_counts() == 1).any())

    def get_one_hot_encoded_node_types(self) -> np.ndarray:
        """bool [nodes, labels]: the labels of every node (all False for an unknown type)."""
        lists = self.get_node_type_ids()
        out = np.zeros((self._n_nodes, len(self.get_unique_node_type_names())), dtype=bool)
        for node, ids in enumerate(lists):
            if ids is not None:
                out[node, ids] = True
        return out

    def has_known_edge_types(self) -> bool:
        return self.has_edge_types() and bool((self.edge_type_ids != UNKNOWN_TYPE).any())

    def get_directed_edges_with_known_edge_types_mask(self) -> np.ndarray:
        return self._edge_types_or_raise() != UNKNOWN_TYPE

    def get_upper_triangular_known_edge_types_mask(self) -> np.ndarray:
        return self.get_directed_edges_with_known_edge_types_mask()[self._upper_triangular()]

    def get_directed_known_edge_type_ids(self) -> np.ndarray:
        ids = self._edge_types_or_raise()
        return ids[ids != UNKNOWN_TYPE]

    def get_upper_triangular_known_edge_type_ids(self) -> np.ndarray:
        ids = self._edge_types_or_raise()[self._upper_triangular()]
        return ids[ids != UNKNOWN_TYPE]

    def get_number_of_known_edge_types(self) -> int:
        """Directed edges whose type is known."""
        return int(self.get_directed_edges_with_known_edge_types_mask().sum())

    def get_edge_type_names_counts_hashmap(self) -> dict:
        """Edge type name -> directed edges of that type (types without edges included)."""
        vocabulary = self.get_unique_edge_type_names()
        counts = np.bincount(self.get_directed_known_edge_type_ids().astype(np.int64),
                             minlength=len(vocabulary))
        return {name: int(c) for name, c in zip(vocabulary, counts)}

    def has_homogeneous_edge_types(self) -> bool:
        return sum(c > 0 for c in self.get_edge_type_names_counts_hashmap().values()) == 1

    def has_singleton_edge_types(self) -> bool:
        """Some type labels exactly one edge (one directed edge, or the two directions of one
        undirected edge)."""
        once = 1 if self._directed else 2
        ids = self._edge_types_or_raise()
        loops = self.get_directed_source_node_ids() == self.col_idx
        per_type = np.bincount(ids[ids != UNKNOWN_TYPE].astype(np.int64))
        loop_per_type = np.bincount(ids[(ids != UNKNOWN_TYPE) & loops].astype(np.int64),
                                    minlength=len(per_type))
        undirected_edges = (per_type + loop_per_type) // once if not self._directed else per_type
        return bool((undirected_edges == 1).any())

    def with_types(self, node_type_ids=None, edge_type_ids=None) -> "CSRGraph":
        """Same graph (arrays shared) with the given type ids attached: u32 arrays, or device
        tensors (int32) for a device-resident graph."""
        if self._device_tensors is not None:
            t = dict(self._device_tensors)
            for key, ids, count in (("node_types", node_type_ids, self._n_nodes),
                                    ("edge_types", edge_type_ids, self._n_edges)):
                if ids is not None:
                    if ids.numel() != count or ids.device != t["row_ptr"].device:
                        raise ValueError(f"{key} must be a device tensor with {count} entries.")
                    t[key] = ids.contiguous()
            g = CSRGraph(None, None, name=self._name, directed=self._directed, _device_tensors=t)
            g._node_names = self._node_names
            return g
        return CSRGraph(self._row_ptr, self._col_idx, self._weights, self._node_names, self._name,
                        self._directed,
                        node_type_ids=self._node_type_ids if node_type_ids is None else node_type_ids,
                        edge_type_ids=self._edge_type_ids if edge_type_ids is None else edge_type_ids,
                        node_type_names=self._node_type_names if node_type_ids is None else None,
                        edge_type_names=self._edge_type_names if edge_type_ids is None else None,
                        node_type_lists=self._node_type_lists if node_type_ids is None else None)

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
        w = None if self._weights is None else self._weights.astype(np.float64)
        et = None if self._edge_type_ids is None else self._edge_type_ids.astype(np.int64)
        row_ptr, cols, w, et = CSRGraph._assemble(
            new_of_old[src_old], new_of_old[self._col_idx.astype(np.int64)], w, et,
            self._n_nodes, True)
        return CSRGraph(
            row_ptr, cols, w, None if names is None else [names[i] for i in order], self._name,
            self._directed,
            node_type_ids=None if self._node_type_ids is None else self._node_type_ids[order],
            edge_type_ids=et, node_type_names=self._node_type_names,
            edge_type_names=self._edge_type_names,
            node_type_lists=(None if self._node_type_lists is None
                             else [self._node_type_lists[i] for i in order]))

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
                         self._directed, node_type_ids=self._node_type_ids,
                         edge_type_ids=self._edge_type_ids,
                         node_type_names=self._node_type_names,
                         edge_type_names=self._edge_type_names,
                         node_type_lists=self._node_type_lists)
        self._degree_normalized = g
        return g

    def content_digest(self) -> str:
        """sha256 over the graph's content (CSR arrays, weights, type ids, node names): what the
        reference's ``@Cache`` hashes when the graph argument takes part in the cache key
        (utils/abstract_models/abstract_embedding_model.py:91-95).  Two graphs with the same name
        but different edges never share a digest."""
        import hashlib

        if getattr(self, "_digest", None) is None:
            h = hashlib.sha256()
            h.update(f"{self._n_nodes},{self._n_edges},{int(self._directed)}".encode())
            for arr in (self.row_ptr, self.col_idx, self.cumw, self.node_type_ids,
                        self.edge_type_ids):
                h.update(b"|")
                if arr is not None:
                    h.update(np.ascontiguousarray(arr).view(np.uint8).data)
            if self._named:
                h.update("\x00".join(map(str, self._node_names)).encode())
            self._digest = h.hexdigest()
        return self._digest

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
    def node_type_ids(self) -> Optional[np.ndarray]:
        if (self._node_type_ids is None and self._device_tensors is not None
                and "node_types" in self._device_tensors):
            self._node_type_ids = self._device_tensors["node_types"].cpu().numpy().view(np.uint32)
        return self._node_type_ids

    @property
    def edge_type_ids(self) -> Optional[np.ndarray]:
        if (self._edge_type_ids is None and self._device_tensors is not None
                and "edge_types" in self._device_tensors):
            self._edge_type_ids = self._device_tensors["edge_types"].cpu().numpy().view(np.uint32)
        return self._edge_type_ids

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

    def _symmetry_flag(self) -> int:
        return 0 if self._directed else _lib.GRAPH_SYMMETRIC

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
                self._n_sources, _lib.GRAPH_DEVICE_PTRS | self._symmetry_flag(), device,
                C.byref(handle)))
            dg = DeviceGraph(handle, device, keep_alive=(t,))
            nt, et = t.get("node_types"), t.get("edge_types")
            if nt is not None or et is not None:
                _lib.check(L.gn2v_graph_set_types(handle, None if nt is None else nt.data_ptr(),
                                                  None if et is None else et.data_ptr()))
        else:
            srcs = self.sources
            _lib.check(L.gn2v_graph_create(
                self._row_ptr.ctypes.data, self._col_idx.ctypes.data,
                None if self._cumw is None else self._cumw.ctypes.data,
                None if srcs is None else srcs.ctypes.data, self._n_nodes, self._n_edges,
                self._n_sources, self._symmetry_flag(), device, C.byref(handle)))
            dg = DeviceGraph(handle, device)
            nt, et = self._node_type_ids, self._edge_type_ids
            if nt is not None or et is not None:
                _lib.check(L.gn2v_graph_set_types(handle, None if nt is None else nt.ctypes.data,
                                                  None if et is None else et.ctypes.data))
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

"""GraphTransformer: a graph (or a list of edges) -> edge embeddings.

Interface and error behaviour of the reference class
(embiggen/embedding_transformers/graph_transformer.py:11-258).  It only decides *which* edges,
node types and edge types are meant (from a graph, an [E, 2] array, a (sources, destinations)
tuple or a list of pairs) and hands them to ``EdgeTransformer``, whose operators run on the GPU.
"""
from typing import List, Union

import numpy as np

from .edge_transformer import EdgeTransformer
from .node_transformer import is_graph


class GraphTransformer:
    def __init__(self, methods: Union[List[str], str] = "Hadamard", aligned_mapping: bool = False,
                 include_both_undirected_edges: bool = True):
        self._transformer = EdgeTransformer(methods=methods, aligned_mapping=aligned_mapping)
        self._include_both_undirected_edges = include_both_undirected_edges
        self._aligned_mapping = aligned_mapping

    def fit(self, node_feature, node_type_feature=None, edge_type_features=None):
        self._transformer.fit(node_feature=node_feature, node_type_feature=node_type_feature,
                              edge_type_features=edge_type_features)
        return self

    def has_node_type_features(self) -> bool:
        return self._transformer.has_node_type_features()

    def has_edge_type_features(self) -> bool:
        return self._transformer.has_edge_type_features()

    def is_aligned_mapping(self) -> bool:
        return self._transformer.is_aligned_mapping()

    def _all_directed(self, graph) -> bool:
        return graph.is_directed() or self._include_both_undirected_edges

    def _endpoints(self, graph):
        """(sources, destinations) of the edges to embed (graph_transformer.py:150-199)."""
        if is_graph(graph):
            if not self._aligned_mapping:
                edges = graph.get_directed_edge_node_names()
            elif self._all_directed(graph):
                edges = (graph.get_directed_source_node_ids(),
                         graph.get_directed_destination_node_ids())
            else:
                edges = (graph.get_source_node_ids(directed=False),
                         graph.get_destination_node_ids(directed=False))
        else:
            edges = graph
        if isinstance(edges, list):
            edges = np.array(edges)
        if isinstance(edges, tuple) and len(edges) == 2 and all(
                isinstance(e, np.ndarray) for e in edges):
            src, dst = edges
            if src.ndim != 1 or dst.ndim != 1 or len(src) == 0 or len(src) != len(dst):
                raise ValueError(
                    "When providing a tuple of numpy arrays containing the source and destination "
                    "node IDs, we expect to receive two arrays both with shape (number of edges,). "
                    f"The ones you have provided have shapes {src.shape} and {dst.shape}.")
            return src, dst
        if isinstance(edges, np.ndarray):
            if edges.ndim != 2 or edges.shape[1] != 2 or edges.shape[0] == 0:
                raise ValueError(
                    "When providing a numpy array containing the source and destination node IDs "
                    "representing the graph edges, we expect to receive an array with shape "
                    f"(number of edges, 2). The one you have provided has shape {edges.shape}.")
            return edges[:, 0], edges[:, 1]
        raise ValueError(
            "The edges must be a graph, an array with shape (number of edges, 2), a tuple of two "
            f"arrays or a list of pairs; got {type(graph)}.")

    def transform(self, graph, node_types=None, edge_types=None, edge_features=None) -> np.ndarray:
        sources, destinations = self._endpoints(graph)
        source_node_types = destination_node_types = None
        if node_types is not None and self.has_node_type_features():
            if is_graph(node_types):
                if self._aligned_mapping:
                    source_node_types = [node_types.get_node_type_ids_from_node_id(v) for v in sources]
                    destination_node_types = [node_types.get_node_type_ids_from_node_id(v)
                                              for v in destinations]
                else:
                    source_node_types = [node_types.get_node_type_names_from_node_name(v)
                                         for v in sources]
                    destination_node_types = [node_types.get_node_type_names_from_node_name(v)
                                              for v in destinations]
            else:
                source_node_types, destination_node_types = node_types
        assert (source_node_types is not None) == self.has_node_type_features()
        if is_graph(edge_types):
            edge_types.must_not_contain_unknown_edge_types()
            edge_types.must_not_be_multigraph()
            if not self.has_edge_type_features():
                raise ValueError(
                    "While the provided graph has edge types, no edge features were provided to "
                    "the graph transformer")
            whole = self._all_directed(edge_types)
            if self.is_aligned_mapping():
                edge_types = (edge_types.get_imputed_directed_edge_type_ids(imputation_edge_type_id=0)
                              if whole else
                              edge_types.get_imputed_upper_triangular_edge_type_ids(
                                  imputation_edge_type_id=0))
            else:
                edge_types = (edge_types.get_directed_edge_type_names() if whole
                              else edge_types.get_upper_triangular_edge_type_names())
        assert (edge_types is not None) == self.has_edge_type_features()
        return self._transformer.transform(
            sources, destinations, source_node_types=source_node_types,
            destination_node_types=destination_node_types, edge_types=edge_types,
            edge_features=edge_features)

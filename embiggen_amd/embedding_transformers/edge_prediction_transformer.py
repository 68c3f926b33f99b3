"""EdgePredictionTransformer: positive and negative edges -> (X, y) for a link-prediction model.

Interface and behaviour of the reference class
(embiggen/embedding_transformers/edge_prediction_transformer.py:10-203): the edges of two graphs
(or edge lists) are embedded by ``GraphTransformer`` -- operators fused with the row gather on the
GPU -- stacked positives first, labelled 1 / 0 and optionally shuffled with a seeded numpy
permutation.
"""
from typing import List, Optional, Tuple, Union

import numpy as np

from .graph_transformer import GraphTransformer
from .node_transformer import is_graph


class EdgePredictionTransformer:
    def __init__(self, methods: Union[List[str], str] = "Hadamard", aligned_mapping: bool = False,
                 include_both_undirected_edges: bool = True):
        self._transformer = GraphTransformer(
            methods=methods, aligned_mapping=aligned_mapping,
            include_both_undirected_edges=include_both_undirected_edges)

    def fit(self, node_feature, node_type_feature=None, edge_type_features=None):
        self._transformer.fit(node_feature, node_type_feature=node_type_feature,
                              edge_type_features=edge_type_features)
        return self

    def _number_of_edges(self, graph) -> int:
        """Rows the transformer makes of `graph` (what edge_features must be aligned with)."""
        return len(self._transformer._endpoints(graph)[0])

    def transform(self, positive_graph, negative_graph,
                  edge_features: Optional[Union[np.ndarray, List[np.ndarray]]] = None,
                  random_state: int = 42, shuffle: bool = False) -> Tuple[np.ndarray, np.ndarray]:
        if is_graph(positive_graph) and is_graph(negative_graph):
            if not positive_graph.has_compatible_node_vocabularies(negative_graph):
                raise ValueError(
                    "The positive and the negative graph are not compatible: their node "
                    "vocabularies differ, or only one of them has node types.")
        if edge_features is None:
            edge_features = []
        if not isinstance(edge_features, list):
            edge_features = [edge_features]
        n_pos = self._number_of_edges(positive_graph)
        n_neg = self._number_of_edges(negative_graph)
        for feature in edge_features:
            if not isinstance(feature, np.ndarray):
                raise ValueError(
                    f"Edge features must be numpy arrays, got {type(feature)}.")
            if feature.shape[0] != n_pos + n_neg:
                raise ValueError(
                    f"Edge features must have one row per edge: got {feature.shape[0]} rows for "
                    f"{n_pos} positive and {n_neg} negative edges.")
        parts = []
        for graph, rows in ((positive_graph, slice(0, n_pos)), (negative_graph, slice(n_pos, None))):
            typed = is_graph(graph)
            parts.append(self._transformer.transform(
                graph,
                node_types=graph if typed and self._transformer.has_node_type_features() else None,
                edge_types=graph if typed and self._transformer.has_edge_type_features() else None,
                edge_features=[f[rows] for f in edge_features] or None))
        x = np.vstack(parts)
        y = np.concatenate([np.ones(parts[0].shape[0]), np.zeros(parts[1].shape[0])])
        if shuffle:
            order = np.random.RandomState(seed=random_state).permutation(y.shape[0])
            x, y = x[order], y[order]
        return x, y

"""EdgeLabelPredictionTransformer: the edges of a typed graph -> (X, edge type labels).

Interface and behaviour of the reference class
(embiggen/embedding_transformers/edge_label_prediction_transformer.py:11-231): checks that the
task makes sense (edge types present, known, more than one, no multigraph), warns about singleton
and unbalanced types, embeds the edges with ``GraphTransformer`` (GPU operators) and returns the
known edge type ids as labels -- booleans when only two types occur; edges of unknown type are
dropped unless ``behaviour_for_unknown_edge_labels="keep"``.
"""
import warnings
from typing import List, Optional, Tuple, Union

import numpy as np

from .graph_transformer import GraphTransformer


class EdgeLabelPredictionTransformer:
    def __init__(self, methods: Union[List[str], str] = "Hadamard", aligned_mapping: bool = False,
                 include_both_undirected_edges: bool = True):
        self._transformer = GraphTransformer(
            methods=methods, aligned_mapping=aligned_mapping,
            include_both_undirected_edges=include_both_undirected_edges)

    def fit(self, node_feature, node_type_feature=None, edge_type_features=None):
        self._transformer.fit(node_feature=node_feature, node_type_feature=node_type_feature,
                              edge_type_features=edge_type_features)
        return self

    def transform(self, graph, edge_features: Optional[Union[np.ndarray, List[np.ndarray]]] = None,
                  behaviour_for_unknown_edge_labels: Optional[str] = None
                  ) -> Tuple[np.ndarray, np.ndarray]:
        if not graph.has_edge_types():
            raise ValueError("Edge-label prediction needs a graph with edge types.")
        if not graph.has_known_edge_types():
            raise ValueError(
                "Edge-label prediction needs known edge types: the graph has an edge type "
                "vocabulary, but no edge carries a type.")
        if graph.has_homogeneous_edge_types():
            raise ValueError(
                "All edges of the graph have the same type: there is nothing to predict.")
        if graph.is_multigraph():
            raise NotImplementedError("Multigraphs are not supported by this class.")
        if graph.has_singleton_edge_types():
            warnings.warn(
                "Some edge type labels a single edge of this graph: predictions for such rare "
                "types are unlikely to generalise.")
        counts = graph.get_edge_type_names_counts_hashmap()
        most_name, most = max(counts.items(), key=lambda item: item[1])
        least_name, least = min(counts.items(), key=lambda item: item[1])
        if most > 20 * least:
            warnings.warn(
                f"Unbalanced edge-label prediction task: edge type `{most_name}` appears {most} "
                f"times, `{least_name}` only {least} times.")
        unknown = graph.has_unknown_edge_types()
        if unknown and behaviour_for_unknown_edge_labels is None:
            warnings.warn(
                "The graph contains edges of unknown type, which will be dropped; choose with "
                "`behaviour_for_unknown_edge_labels` ('drop' or 'keep') to silence this warning.")
            behaviour_for_unknown_edge_labels = "drop"
        x = self._transformer.transform(graph, node_types=graph, edge_features=edge_features)
        # the rows GraphTransformer produced: every directed edge, or the upper triangle
        whole = x.shape[0] == graph.get_number_of_directed_edges()
        labels = (graph.get_directed_known_edge_type_ids() if whole
                  else graph.get_upper_triangular_known_edge_type_ids())
        if sum(c > 0 for c in counts.values()) == 2:
            labels = labels == 1  # as the reference: the type of id 1 is the positive class
        if unknown and behaviour_for_unknown_edge_labels == "drop":
            x = x[graph.get_directed_edges_with_known_edge_types_mask() if whole
                  else graph.get_upper_triangular_known_edge_types_mask()]
        if labels.size != x.shape[0]:
            raise ValueError(
                f"{x.shape[0]} edge embeddings for {labels.size} edge labels: edges of unknown "
                "type can only be dropped (`behaviour_for_unknown_edge_labels='drop'`).")
        return x, labels

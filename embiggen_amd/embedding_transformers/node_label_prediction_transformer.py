"""NodeLabelPredictionTransformer: the nodes of a typed graph -> (X, node type labels).

Interface and behaviour of the reference class
(embiggen/embedding_transformers/node_label_prediction_transformer.py:11-152): node features
through ``NodeTransformer``, labels = the node type id (float, NaN when unknown) or the one-hot
matrix of a multi-label graph; nodes of unknown type are dropped by default, with a warning when
the caller did not choose.
"""
import warnings
from typing import Optional, Tuple

import numpy as np

from .node_transformer import NodeTransformer


class NodeLabelPredictionTransformer:
    def __init__(self, aligned_mapping: bool = False):
        self._transformer = NodeTransformer(aligned_mapping=aligned_mapping)

    def fit(self, node_feature):
        self._transformer.fit(node_feature)
        return self

    def transform(self, graph, behaviour_for_unknown_node_labels: Optional[str] = "warn",
                  shuffle: bool = False, random_state: int = 42) -> Tuple[np.ndarray, np.ndarray]:
        name = graph.get_name()
        if not graph.has_node_types():
            raise ValueError(f"Node-label prediction needs node types; graph {name} has none.")
        if not graph.has_known_node_types():
            raise ValueError(
                f"Node-label prediction needs known node types: graph {name} has a node type "
                "vocabulary, but no node carries a type.")
        if graph.has_homogeneous_node_types():
            raise ValueError(
                f"All nodes of graph {name} have the same type: there is nothing to predict.")
        if graph.has_singleton_node_types():
            warnings.warn(
                f"Some node type labels a single node of graph {name}: predictions for such rare "
                "types are unlikely to generalise.")
        unknown = graph.has_unknown_node_types()
        if unknown and behaviour_for_unknown_node_labels == "warn":
            warnings.warn(
                "The graph contains nodes of unknown type, which will be dropped; choose with "
                "`behaviour_for_unknown_node_labels` ('drop' or 'keep') to silence this warning.")
            behaviour_for_unknown_node_labels = "drop"
        x = self._transformer.transform(graph)
        if graph.has_multilabel_node_types():
            y = graph.get_one_hot_encoded_node_types()
        else:
            y = np.array([np.nan if ids is None else ids[0] for ids in graph.get_node_type_ids()],
                         dtype=np.float32)
        if unknown and behaviour_for_unknown_node_labels == "drop":
            known = graph.get_nodes_with_known_node_types_mask()
            x, y = x[known], y[known]
        if shuffle:
            order = np.random.RandomState(seed=random_state).permutation(x.shape[0])
            x, y = x[order], y[order]
        return x, y

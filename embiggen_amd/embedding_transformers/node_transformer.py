"""NodeTransformer: node (and node type) features -> per-node feature rows.

Interface and error behaviour of the reference class
(embiggen/embedding_transformers/node_transformer.py:8-243): ``fit`` takes numpy arrays / pandas
DataFrames (or lists of them, stacked column-wise, :110-130), ``transform`` takes node ids (aligned
mapping) or node names (DataFrame index) or a graph.  On top of that the fitted node features are
available as one float32 table in HBM (``device_table``) with ``positions`` mapping ids / names to
its rows: ``EdgeTransformer`` feeds both to the fused gather + operator kernel, so the gathered
source / destination matrices the reference builds with numpy never exist.

Deliberate difference: with ``aligned_mapping=False`` the reference looks node type names up in the
*node* feature frame (node_transformer.py:222, a slip: the node type frame is meant); this class
uses the node type feature frame.
"""
from typing import Optional

import numpy as np
import pandas as pd


def is_graph(obj) -> bool:
    """ensmallen.Graph or our CSRGraph (the reference tests ``isinstance(x, Graph)``)."""
    return hasattr(obj, "get_directed_destination_node_ids") and hasattr(obj, "get_node_names")


def _as_list(features) -> list:
    if features is None:
        return []
    return features if isinstance(features, list) else [features]


def _validate(features: list, kind: str, aligned: bool):
    for feature in features:
        if not isinstance(feature, (pd.DataFrame, np.ndarray)):
            raise ValueError(
                f"One of the provided {kind} features is neither a pandas DataFrame nor a numpy "
                f"array, but of type {type(feature)}. It is not clear what to do with this feature."
            )
        values = feature.to_numpy() if isinstance(feature, pd.DataFrame) else feature
        if pd.isna(values).any():
            raise ValueError(
                f"One of the provided {kind} features contains NaNs. This is not supported. "
                f"The feature has shape {feature.shape}."
            )
    if not aligned and any(isinstance(f, np.ndarray) for f in features):
        raise ValueError(
            "A numpy array feature was provided while the aligned mapping parameter was set to "
            f"false. If you intend to specify that you are providing a numpy array {kind} feature "
            "that is aligned with the vocabulary of the graph set the `aligned_mapping` parameter "
            "to True."
        )


def _stack(features: list, aligned: bool):
    """[] | ndarray (aligned) | DataFrame (named rows)."""
    if not features:
        return []
    if aligned:
        arrays = [f.to_numpy() if isinstance(f, pd.DataFrame) else f for f in features]
        return arrays[0] if len(arrays) == 1 else np.hstack(arrays)
    return pd.concat(features, axis=1)


class NodeTransformer:
    def __init__(self, aligned_mapping: bool = False):
        self._node_feature = []
        self._node_type_feature = []
        self._aligned_mapping = aligned_mapping
        self._device_table = None

    def fit(self, node_feature=None, node_type_feature=None):
        node_feature, node_type_feature = _as_list(node_feature), _as_list(node_type_feature)
        _validate(node_type_feature, "node type", self._aligned_mapping)
        _validate(node_feature, "node", self._aligned_mapping)
        self._node_feature = _stack(node_feature, self._aligned_mapping)
        self._node_type_feature = _stack(node_type_feature, self._aligned_mapping)
        self._device_table = None

    def has_node_type_features(self) -> bool:
        return len(self._node_type_feature) > 0

    def has_node_features(self) -> bool:
        return len(self._node_feature) > 0

    def is_aligned_mapping(self) -> bool:
        return self._aligned_mapping

    def is_fit(self) -> bool:
        return len(self._node_feature) + len(self._node_type_feature) > 0

    # ------------------------------------------------------------------ row lookup
    def positions(self, nodes) -> Optional[np.ndarray]:
        """Rows of the node feature table for ``nodes``: None = every row in order (a graph under
        aligned mapping), else an int64 array.  Same acceptance rules as the reference's
        ``transform`` (:179-190, :210-214)."""
        if self._aligned_mapping:
            if is_graph(nodes):
                return None
            if not isinstance(nodes, np.ndarray):
                raise ValueError(
                    "The provided nodes are not numpy array and the node IDs or Graph are "
                    "expected to be aligned."
                )
            ids = nodes.astype(np.int64)
            if len(ids) and (ids.min() < 0 or ids.max() >= len(self._node_feature)):
                raise IndexError("Node id outside the fitted node features.")
            return ids
        if is_graph(nodes):
            nodes = nodes.get_node_names()
        found = self._node_feature.index.get_indexer(list(nodes))
        if (found < 0).any():
            missing = [n for n, f in zip(nodes, found) if f < 0]
            raise KeyError(f"{missing[:5]} not in the index of the node features.")
        return found.astype(np.int64)

    def _node_values(self) -> np.ndarray:
        f = self._node_feature
        return f.to_numpy() if isinstance(f, pd.DataFrame) else f

    def device_table(self, device: int = 0):
        """The fitted node features as one contiguous float32 [N, d] tensor in HBM (cached)."""
        import torch

        if self._device_table is None or self._device_table.device.index != device:
            values = np.ascontiguousarray(self._node_values(), dtype=np.float32)
            self._device_table = torch.from_numpy(values).to(torch.device("cuda", device))
        return self._device_table

    def node_type_rows(self, nodes, node_types) -> np.ndarray:
        """[n, dt] mean node type feature of every node (zeros for unknown), :192-206."""
        table = self._node_type_feature
        width = table.shape[1]
        if is_graph(node_types):
            per_node = (node_types.get_node_type_ids() if self._aligned_mapping
                        else node_types.get_node_type_names())
            if isinstance(nodes, np.ndarray):
                per_node = [per_node[int(v)] for v in nodes]
            elif nodes is not None and not is_graph(nodes):
                per_node = [per_node[node_types.get_node_id_from_node_name(v)] for v in nodes]
            node_types = per_node
        rows = []
        for own in node_types:
            if own is None:
                rows.append(np.zeros(width))
            elif self._aligned_mapping:
                rows.append(np.mean(table[np.asarray(own, dtype=np.int64)], axis=0))
            else:
                rows.append(np.mean(table.loc[list(own)].to_numpy(), axis=0))
        return np.vstack(rows)

    def transform(self, nodes=None, node_types=None) -> np.ndarray:
        if not self.is_fit():
            raise ValueError("Transformer was not fitted yet.")
        node_features = node_type_features = None
        if nodes is not None and self.has_node_features():
            rows = self.positions(nodes)
            values = self._node_values()
            node_features = values if rows is None else values[rows]
        if node_types is not None and self.has_node_type_features():
            node_type_features = self.node_type_rows(nodes, node_types)
        if node_features is None:
            return node_type_features
        if node_type_features is None:
            return node_features
        return np.hstack([node_features, node_type_features])

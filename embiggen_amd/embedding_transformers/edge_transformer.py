"""Node-embedding -> edge-embedding operators on the GPU.

Device counterpart of the reference's ``EdgeTransformer`` operators
(embiggen/embedding_transformers/edge_transformer.py:12-343; method table :348-361): the same
method names and outputs, but computed by one fused gather + operator kernel
(``gn2v_edge_embedding``) straight from the embedding table(s) in HBM instead of materialising the
gathered source / destination matrices with numpy.  This is the step right after the embedder in
the reference's classifier pipelines (abstract_classifier_model.py:717-757).
"""
from typing import List, Optional, Union

import numpy as np
import pandas as pd

from .. import _lib
from .node_transformer import NodeTransformer

METHODS = {
    "Hadamard": 0, "Sum": 1, "Average": 2, "L1": 3, "AbsoluteL1": 4, "SquaredL2": 5, "L2": 6,
    "Concatenate": 7, "Min": 8, "Max": 9, "L2Distance": 10, "CosineSimilarity": 11,
}


def _output_width(method: str, d: int) -> int:
    if method == "Concatenate":
        return 2 * d
    if method in ("L2Distance", "CosineSimilarity"):
        return 1
    return d


def edge_embedding(table, sources, destinations, method: str = "Hadamard",
                   destination_table=None, embedding_size: Optional[int] = None):
    """out[e] = method(table[sources[e]], destination_table[destinations[e]]) as a float32 CUDA
    tensor.  ``table`` is [N, ld] float32 on the device (``embedding_size`` = number of real
    columns when the rows are padded); ids are int32 / int64 tensors on the same device."""
    import torch

    if method not in METHODS:
        raise ValueError(
            f"The provided edge embedding method `{method}` is not supported. "
            f"The supported methods are {list(METHODS)}."
        )
    dst_table = table if destination_table is None else destination_table
    if table.dtype != torch.float32 or dst_table.dtype != torch.float32:
        raise ValueError("Embedding tables must be float32.")
    if table.shape[1] != dst_table.shape[1]:
        raise ValueError("Source and destination tables must have the same row stride.")
    if sources.shape != destinations.shape:
        raise ValueError("sources and destinations must have the same shape.")
    ld = table.shape[1]
    d = ld if embedding_size is None else embedding_size
    pad = (-ld) % 4
    if pad:  # the kernel reads 16 B chunks: pad odd strides once
        table = torch.nn.functional.pad(table, (0, pad))
        dst_table = table if destination_table is None else torch.nn.functional.pad(dst_table,
                                                                                    (0, pad))
        ld += pad
    table, dst_table = table.contiguous(), dst_table.contiguous()
    src = sources.to(torch.int32).contiguous()
    dst = destinations.to(torch.int32).contiguous()
    width = _output_width(method, d)
    out = torch.empty((src.numel(), width), dtype=torch.float32, device=table.device)
    stream = torch.cuda.current_stream(table.device).cuda_stream
    _lib.check(_lib.lib().gn2v_edge_embedding(
        table.data_ptr(), dst_table.data_ptr(), d, ld, src.data_ptr(), dst.data_ptr(),
        src.numel(), METHODS[method], out.data_ptr(), width, stream))
    return out


_INT_TYPES = (int, np.integer)


def _is_tensor(x) -> bool:
    return type(x).__module__.startswith("torch") and hasattr(x, "data_ptr")


class EdgeTransformer:
    """Edges -> edge embeddings.  Interface and error behaviour of the reference class
    (edge_transformer.py:345-718): ``methods`` (one or more of the 12 operators; several are
    concatenated column-wise), ``aligned_mapping``, ``fit(node_feature, node_type_feature,
    edge_type_features)`` on numpy arrays / DataFrames, ``transform(sources, destinations, ...)``
    returning ``[edge embeddings | edge features | edge type features]`` as a numpy array.

    The operators run in the fused gather + operator kernel on the node feature table kept in HBM
    (float32); only ids go to the device and only the result comes back.  Device consumers can
    skip the host altogether: ``fit`` on float32 CUDA tensor(s) and ``transform`` on id tensors
    returns a CUDA tensor.

    Deliberate difference: the reference's ``fit`` raises ``ValueError`` for *every* DataFrame
    edge type feature (``if pd.isna(df).any()`` at edge_transformer.py:472 is the truth value of
    a Series); DataFrame edge type features work here as its docstring (:454-459) intends."""

    methods = METHODS

    def __init__(self, methods: Union[str, List[str]] = "Hadamard", aligned_mapping: bool = False):
        if not isinstance(methods, list):
            methods = [methods]
        for m in methods:
            if not isinstance(m, str):
                raise ValueError(
                    f"The provided method name should be a string, but we got {type(m)} instead.")
            if m not in METHODS:
                raise ValueError(
                    f"The provided edge embedding method `{m}` is not supported. "
                    f"The supported edge embedding methods are {list(METHODS)}.")
        self._method_names = list(methods)
        self._methods = self._method_names
        self._transformer = NodeTransformer(aligned_mapping=aligned_mapping)
        self._edge_type_features = []
        self._table = None  # device-tensor mode

    # ------------------------------------------------------------------ fit
    def fit(self, node_feature, node_type_feature=None, edge_type_features=None):
        tensors = node_feature if isinstance(node_feature, (list, tuple)) else [node_feature]
        if len(tensors) and all(_is_tensor(t) for t in tensors):
            # device mode: float32 CUDA tensor(s), hstacked like node_transformer.py:110-116
            import torch

            if node_type_feature is not None or edge_type_features is not None:
                raise ValueError("Device tensors can only be fitted as node features.")
            self._table = (tensors[0] if len(tensors) == 1 else torch.cat(list(tensors), dim=1)).contiguous()
            return self
        self._table = None
        if edge_type_features is None:
            edge_type_features = []
        if not isinstance(edge_type_features, list):
            edge_type_features = [edge_type_features]
        for feature in edge_type_features:
            if not isinstance(feature, (pd.DataFrame, np.ndarray)):
                raise ValueError(
                    "The provided edge type features should be either Pandas Dataframes or "
                    f"Numpy arrays, but we got {type(feature)} instead.")
            if isinstance(feature, pd.DataFrame):
                if feature.index.hasnans:
                    raise ValueError(
                        "The provided edge type features should not have NaN values in their index.")
                if feature.index.has_duplicates:
                    raise ValueError(
                        "The provided edge type features should not have duplicated values in "
                        "their index.")
            values = feature.to_numpy() if isinstance(feature, pd.DataFrame) else feature
            if pd.isna(values).any():
                raise ValueError(
                    "The provided edge type features should not have NaN values in their values.")
        self._edge_type_features = edge_type_features
        self._transformer.fit(node_feature=node_feature, node_type_feature=node_type_feature)
        return self

    def has_edge_type_features(self) -> bool:
        return len(self._edge_type_features) > 0

    def has_node_type_features(self) -> bool:
        return self._transformer.has_node_type_features()

    def is_aligned_mapping(self) -> bool:
        return self._transformer.is_aligned_mapping()

    def has_numpy_edge_type_features(self) -> bool:
        return any(isinstance(f, np.ndarray) for f in self._edge_type_features)

    # ------------------------------------------------------------------ transform
    def _edge_type_rows(self, edge_types) -> List[np.ndarray]:
        rows = []
        for feature in self._edge_type_features:
            first = edge_types[0]
            if isinstance(first, str):
                if isinstance(feature, np.ndarray):
                    raise ValueError(
                        "Since the edge type features are provided as numpy arrays, the edge types "
                        f"should be provided as integers. We got instead {type(first)}.")
                rows.append(feature.loc[list(edge_types)].values)
            elif isinstance(first, _INT_TYPES) and not isinstance(first, bool):
                ids = np.asarray(edge_types, dtype=np.int64)
                rows.append(feature.iloc[ids].values if isinstance(feature, pd.DataFrame)
                            else feature[ids])
            else:
                raise ValueError(
                    "The provided edge types should be either strings or integers, but we got "
                    f"{type(first)} instead.")
        return rows

    def _embeddings(self, sources, destinations, source_node_types, destination_node_types,
                    device: int = 0) -> List[np.ndarray]:
        """One float32 array per method, computed on the device."""
        import torch

        nt = self._transformer
        table = nt.device_table(device) if nt.has_node_features() else None
        dev = torch.device("cuda", device)

        def ids_of(nodes):
            rows = nt.positions(nodes)
            if rows is None:
                rows = np.arange(table.shape[0], dtype=np.int64)
            return torch.from_numpy(np.ascontiguousarray(rows)).to(dev)

        if not nt.has_node_type_features():
            src, dst = ids_of(sources), ids_of(destinations)
            src_table = dst_table = table
        else:
            # [node features | mean node type features] per endpoint, assembled in HBM
            sides = []
            for nodes, types in ((sources, source_node_types), (destinations, destination_node_types)):
                parts = []
                if table is not None:
                    parts.append(table.index_select(0, ids_of(nodes)))
                if types is not None:
                    rows = np.ascontiguousarray(nt.node_type_rows(nodes, types), dtype=np.float32)
                    parts.append(torch.from_numpy(rows).to(dev))
                sides.append(parts[0] if len(parts) == 1 else torch.cat(parts, dim=1))
            src_table, dst_table = sides
            if src_table.shape != dst_table.shape:
                raise ValueError("Source and destination features have different shapes.")
            src = dst = torch.arange(src_table.shape[0], dtype=torch.int64, device=dev)
        out = []
        for method in self._method_names:
            emb = edge_embedding(src_table, src, dst, method, destination_table=dst_table)
            assert not bool(torch.isnan(emb).any()), (
                "The provided edge embedding should not have NaN values, but we got a tensor with "
                f"shape {tuple(emb.shape)} and NaN values. The object was obtained using the "
                f"method {method}.")
            out.append(emb.cpu().numpy())
        return out

    def transform(self, sources, destinations, source_node_types=None,
                  destination_node_types=None, edge_types=None, edge_features=None):
        if self._table is not None:
            import torch

            parts = [edge_embedding(self._table, sources, destinations, m)
                     for m in self._method_names]
            return parts[0] if len(parts) == 1 else torch.cat(parts, dim=1)
        if self.has_edge_type_features() and edge_types is None:
            raise ValueError("The edge type features are provided, but no edge types are provided.")
        if self.has_node_type_features() and source_node_types is None:
            raise ValueError(
                "The node type features are provided, but no source node types are provided.")
        if len(sources) != len(destinations):
            raise ValueError(
                "The provided sources and destinations should have the same length, "
                f"but we got {len(sources)} and {len(destinations)} instead.")
        if edge_types is not None and len(destinations) != len(edge_types):
            raise ValueError(
                "The provided sources, destinations and edge types should have the same length, "
                f"but we got {len(sources)}, {len(destinations)} and {len(edge_types)} instead.")
        edge_type_features = self._edge_type_rows(edge_types) if self.has_edge_type_features() else []
        edge_embeddings = []
        if self._transformer.is_fit():
            edge_embeddings = self._embeddings(sources, destinations, source_node_types,
                                               destination_node_types)
        if edge_features is None:
            edge_features = []
        if not isinstance(edge_features, list):
            edge_features = [edge_features]
        for feature in edge_features:
            if not isinstance(feature, np.ndarray):
                raise ValueError(
                    f"The provided edge features should be numpy arrays, but we got {type(feature)} "
                    "instead.")
        if not (edge_features or edge_type_features or edge_embeddings):
            raise ValueError(
                "At least one of the provided edge features, edge type features or edge "
                "embeddings should be provided.")
        blocks = edge_embeddings + edge_features + edge_type_features
        expected = blocks[0].shape[0] if edge_embeddings else (
            edge_features[0].shape[0] if edge_features else edge_type_features[0].shape[0])
        for block in blocks:
            assert not pd.isna(block).any(), (
                "The provided edge features should not have NaN values, but we got a numpy array "
                f"with shape {block.shape} and NaN values.")
            if block.shape[0] != expected:
                raise ValueError(
                    "The provided edge features should have a sample for each of the edges in the "
                    f"graph, which are {expected}, but we got {block.shape[0]}.")
        return np.hstack([block.reshape((expected, -1)) for block in blocks])

"""Node-embedding -> edge-embedding operators on the GPU.

Device counterpart of the reference's ``EdgeTransformer`` operators
(embiggen/embedding_transformers/edge_transformer.py:12-343; method table :348-361): the same
method names and outputs, but computed by one fused gather + operator kernel
(``gn2v_edge_embedding``) straight from the embedding table(s) in HBM instead of materialising the
gathered source / destination matrices with numpy.  This is the step right after the embedder in
the reference's classifier pipelines (abstract_classifier_model.py:717-757).
"""
from typing import List, Optional, Union

from .. import _lib

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


class EdgeTransformer:
    """Minimal device version of the reference class (edge_transformer.py:345-): one or more
    methods, ``fit`` on the node feature table(s), ``transform`` on (sources, destinations); the
    outputs of several methods are concatenated column-wise as in the reference (:…transform)."""

    methods = METHODS

    def __init__(self, methods: Union[str, List[str]] = "Hadamard"):
        if isinstance(methods, str):
            methods = [methods]
        for m in methods:
            if m not in METHODS:
                raise ValueError(f"Unknown edge embedding method `{m}`; supported: {list(METHODS)}.")
        self._methods = list(methods)
        self._table = None

    def fit(self, node_feature):
        """node_feature: float32 CUDA tensor [N, d], or a list of them (hstacked like the
        reference's NodeTransformer, node_transformer.py:110)."""
        import torch

        if isinstance(node_feature, (list, tuple)):
            node_feature = torch.cat(list(node_feature), dim=1)
        self._table = node_feature.contiguous()
        return self

    def transform(self, sources, destinations):
        import torch

        if self._table is None:
            raise ValueError("Transformer was not fitted yet.")
        parts = [edge_embedding(self._table, sources, destinations, m) for m in self._methods]
        return parts[0] if len(parts) == 1 else torch.cat(parts, dim=1)

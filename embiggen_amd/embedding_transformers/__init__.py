"""Device-side embedding transformers (drop-in names of ``embiggen.embedding_transformers``)."""
from .edge_transformer import METHODS, EdgeTransformer, edge_embedding
from .graph_transformer import GraphTransformer
from .node_transformer import NodeTransformer

__all__ = ["EdgeTransformer", "NodeTransformer", "GraphTransformer", "edge_embedding", "METHODS"]

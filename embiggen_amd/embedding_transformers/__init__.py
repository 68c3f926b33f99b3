"""Device-side embedding transformers (drop-in names of ``embiggen.embedding_transformers``)."""
from .edge_transformer import METHODS, EdgeTransformer, edge_embedding

__all__ = ["EdgeTransformer", "edge_embedding", "METHODS"]

"""Device-side embedding transformers (drop-in names of ``embiggen.embedding_transformers``)."""
from .edge_label_prediction_transformer import EdgeLabelPredictionTransformer
from .edge_prediction_transformer import EdgePredictionTransformer
from .edge_transformer import METHODS, EdgeTransformer, edge_embedding
from .graph_transformer import GraphTransformer
from .node_label_prediction_transformer import NodeLabelPredictionTransformer
from .node_transformer import NodeTransformer

__all__ = ["EdgeTransformer", "NodeTransformer", "GraphTransformer", "EdgePredictionTransformer",
           "EdgeLabelPredictionTransformer", "NodeLabelPredictionTransformer", "edge_embedding",
           "METHODS"]

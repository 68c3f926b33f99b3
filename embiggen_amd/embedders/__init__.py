"""Node embedding models of the gn2v engine (drop-in names of ``embiggen.embedders``)."""
from .graph_embedding_pipeline import embed_graph
from .node2vec import (DeepWalkCBOWEnsmallen, DeepWalkGloVeEnsmallen, DeepWalkSkipGramEnsmallen,
                       EnsmallenEmbedder, Node2VecCBOWEnsmallen, Node2VecEnsmallen,
                       Node2VecGloVeEnsmallen, Node2VecSkipGramEnsmallen, WalkletsCBOWEnsmallen,
                       WalkletsEnsmallen, WalkletsGloVeEnsmallen, WalkletsSkipGramEnsmallen)

__all__ = [
    "embed_graph", "EnsmallenEmbedder", "Node2VecEnsmallen", "Node2VecSkipGramEnsmallen",
    "Node2VecCBOWEnsmallen", "DeepWalkSkipGramEnsmallen", "DeepWalkCBOWEnsmallen",
    "WalkletsEnsmallen", "WalkletsSkipGramEnsmallen", "WalkletsCBOWEnsmallen",
    "Node2VecGloVeEnsmallen", "DeepWalkGloVeEnsmallen", "WalkletsGloVeEnsmallen",
]

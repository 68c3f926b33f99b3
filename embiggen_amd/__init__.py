"""embiggen_amd -- MI355X-native Node2Vec / SkipGram / CBOW engine behind embiggen's embedder API.

Scope: the one data-parallel hot path of embiggen, i.e.
``Node2Vec{SkipGram,CBOW}Ensmallen(...).fit_transform(graph) -> EmbeddingResult`` and the walk /
window batch generator of ``embiggen.sequences``.  Compute runs in hand-written HIP kernels for
gfx950 reached through the C ABI in ``include/gn2v.h``; there is no CPU execution path.
"""
from . import _lib
from .embedding_transformers import (EdgeLabelPredictionTransformer, EdgePredictionTransformer,
                                     EdgeTransformer, GraphTransformer,
                                     NodeLabelPredictionTransformer, NodeTransformer)
from .embedders import (DeepWalkCBOWEnsmallen, DeepWalkGloVeEnsmallen, DeepWalkSkipGramEnsmallen,
                        Node2VecCBOWEnsmallen, Node2VecGloVeEnsmallen, Node2VecSkipGramEnsmallen,
                        WalkletsCBOWEnsmallen, WalkletsGloVeEnsmallen,
                        WalkletsSkipGramEnsmallen, embed_graph)
from .graph import CSRGraph, barabasi_albert, karate_club
from .sequences import Node2VecSequence
from .utils import (AbstractEmbeddingModel, AbstractModel, EmbeddingResult,
                    get_available_models_for_node_embedding, get_models_dataframe,
                    normalize_kwargs)

__version__ = "0.1.0"

__all__ = [
    "CSRGraph", "karate_club", "barabasi_albert", "EmbeddingResult", "AbstractModel",
    "AbstractEmbeddingModel", "embed_graph", "Node2VecSkipGramEnsmallen",
    "Node2VecCBOWEnsmallen", "DeepWalkSkipGramEnsmallen", "DeepWalkCBOWEnsmallen",
    "WalkletsSkipGramEnsmallen", "WalkletsCBOWEnsmallen", "Node2VecGloVeEnsmallen",
    "DeepWalkGloVeEnsmallen", "WalkletsGloVeEnsmallen",
    "get_models_dataframe", "get_available_models_for_node_embedding", "normalize_kwargs",
    "Node2VecSequence", "EdgeTransformer", "NodeTransformer", "GraphTransformer",
    "EdgePredictionTransformer", "EdgeLabelPredictionTransformer",
    "NodeLabelPredictionTransformer",
]

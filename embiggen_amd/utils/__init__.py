from .abstract_model import (AbstractEmbeddingModel, AbstractModel, abstract_class,
                             get_available_models_for_node_embedding, get_models_dataframe)
from .embedding_result import EmbeddingResult
from .normalize_kwargs import normalize_kwargs

__all__ = [
    "AbstractModel", "AbstractEmbeddingModel", "abstract_class", "EmbeddingResult",
    "normalize_kwargs", "get_models_dataframe", "get_available_models_for_node_embedding",
]

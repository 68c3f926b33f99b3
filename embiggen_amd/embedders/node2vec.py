"""Node2Vec / DeepWalk SkipGram & CBOW embedders with the reference's class names and kwargs.

Drop-in for embiggen/embedders/ensmallen_embedders/{node2vec.py, node2vec_skipgram.py,
node2vec_cbow.py, deepwalk_skipgram.py, deepwalk_cbow.py, ensmallen_embedder.py}: same
constructor signatures and defaults (node2vec_skipgram.py:9-36), ``parameters()`` round trip with
the removed keys (:148-161), ``smoke_test_parameters()`` (node2vec.py:79-87), capability flags
(node2vec.py:114-166, ensmallen_embedder.py:41-55) and result order -- SkipGram
``[central, contextual]``, CBOW reversed so the input-side table is first (node2vec.py:101-102).
The compute call ``self._model.fit_transform(graph)`` (node2vec.py:99) lands in
``embiggen_amd.models`` -> ``libgn2v.so`` instead of the ensmallen wheel.
"""
from typing import Any, Dict, Optional

import pandas as pd

from .. import models
from ..utils.abstract_model import AbstractEmbeddingModel, abstract_class, must_be_in_set
from ..utils.embedding_result import EmbeddingResult
from ..utils.normalize_kwargs import normalize_kwargs


@abstract_class
class EnsmallenEmbedder(AbstractEmbeddingModel):
    """Base of the embedders that the reference backs with the ensmallen wheel."""

    def __init__(self, random_state: Optional[int] = None, embedding_size: Optional[int] = None,
                 ring_bell: bool = False, enable_cache: bool = False):
        super().__init__(random_state=random_state, embedding_size=embedding_size,
                         ring_bell=ring_bell, enable_cache=enable_cache)

    @classmethod
    def task_name(cls) -> str:
        return "Node Embedding"

    @classmethod
    def library_name(cls) -> str:
        return "Ensmallen"

    @classmethod
    def requires_nodes_sorted_by_decreasing_node_degree(cls) -> bool:
        return False

    @classmethod
    def is_topological(cls) -> bool:
        return True


@abstract_class
class Node2VecEnsmallen(EnsmallenEmbedder):
    """Abstract walk-based embedder: picks the engine model by ``model_name()``."""

    MODELS = {
        "DeepWalk CBOW": models.CBOW,
        "DeepWalk SkipGram": models.SkipGram,
        "Node2Vec CBOW": models.CBOW,
        "Node2Vec SkipGram": models.SkipGram,
        "Walklets CBOW": models.WalkletsCBOW,
        "Walklets SkipGram": models.WalkletsSkipGram,
        "DeepWalk GloVe": models.GloVe,
        "Node2Vec GloVe": models.GloVe,
        "Walklets GloVe": models.WalkletsGloVe,
    }

    def __init__(self, embedding_size: int = 100, random_state: int = 42,
                 ring_bell: bool = False, enable_cache: bool = False, **model_kwargs: Dict):
        model_name = must_be_in_set(self.model_name(), self.MODELS.keys(), "model name")
        self._model_kwargs = normalize_kwargs(
            self, {**model_kwargs, "embedding_size": embedding_size, "random_state": random_state}
        )
        embedding_size = self._model_kwargs.pop("embedding_size")
        random_state = self._model_kwargs.pop("random_state")
        self._model = Node2VecEnsmallen.MODELS[model_name](
            embedding_size=embedding_size, random_state=random_state, **self._model_kwargs
        )
        super().__init__(embedding_size=embedding_size, enable_cache=enable_cache,
                         ring_bell=ring_bell, random_state=random_state)

    @classmethod
    def smoke_test_parameters(cls) -> Dict[str, Any]:
        return dict(epochs=1, embedding_size=5, window_size=1, walk_length=4, max_neighbours=10)

    _REMOVED_PARAMETERS = ("change_node_type_weight", "change_edge_type_weight", "alpha")

    def parameters(self) -> Dict[str, Any]:
        merged = dict(**super().parameters(), **self._model_kwargs)
        return {k: v for k, v in merged.items() if k not in self._REMOVED_PARAMETERS}

    def set_random_state(self, random_state: int):
        super().set_random_state(random_state)
        self._model.random_state = int(random_state)

    def _fit_transform(self, graph, return_dataframe: bool = True) -> EmbeddingResult:
        node_embeddings = self._model.fit_transform(graph)
        if "CBOW" in self.model_name():
            node_embeddings = list(reversed(node_embeddings))
        if return_dataframe:
            node_names = graph.get_node_names()
            node_embeddings = [
                pd.DataFrame(embedding, index=node_names) for embedding in node_embeddings
            ]
        return EmbeddingResult(embedding_method_name=self.model_name(),
                               node_embeddings=node_embeddings)

    def get_last_stats(self) -> Optional[Dict[str, Any]]:
        """Counters / kernel times of the last fit (engine extension, not in the reference)."""
        return self._model.last_stats

    def set_distributed(self, comm) -> "Node2VecEnsmallen":
        """Opt in to multi-GPU training (engine extension, not in the reference): ``comm`` is an
        ``embiggen_amd.distributed.TorchComm`` over the job's process group (one process per
        GPU); every rank must then make the same ``fit_transform`` call and receives the full
        tables.  ``None`` switches back to this process's own device.  Models without a
        multi-GPU path (CBOW, GloVe) ignore it."""
        self._model.comm = comm
        return self

    @classmethod
    def requires_edge_weights(cls) -> bool:
        return False

    @classmethod
    def requires_positive_edge_weights(cls) -> bool:
        return True

    @classmethod
    def can_use_edge_weights(cls) -> bool:
        return True

    def is_using_edge_weights(self) -> bool:
        return True

    @classmethod
    def can_use_node_types(cls) -> bool:
        return True

    def is_using_node_types(self) -> bool:
        return self._model_kwargs.get("change_node_type_weight", 1.0) != 1.0

    @classmethod
    def can_use_edge_types(cls) -> bool:
        return True

    def is_using_edge_types(self) -> bool:
        return self._model_kwargs.get("change_edge_type_weight", 1.0) != 1.0

    @classmethod
    def is_stocastic(cls) -> bool:
        return True

    @classmethod
    def requires_node_types(cls) -> bool:
        return False

    @classmethod
    def requires_edge_types(cls) -> bool:
        return False


def _forward(local_vars: Dict[str, Any]) -> Dict[str, Any]:
    return {k: v for k, v in local_vars.items() if k not in ("self", "__class__")}


class Node2VecSkipGramEnsmallen(Node2VecEnsmallen):
    """Node2Vec SkipGram on the MI355X engine (reference: node2vec_skipgram.py:6-166)."""

    def __init__(
        self,
        embedding_size: int = 100,
        epochs: int = 30,
        clipping_value: float = 6.0,
        number_of_negative_samples: int = 10,
        walk_length: int = 128,
        iterations: int = 10,
        window_size: int = 5,
        return_weight: float = 0.25,
        explore_weight: float = 4.0,
        change_node_type_weight: float = 1.0,
        change_edge_type_weight: float = 1.0,
        max_neighbours: Optional[int] = 100,
        learning_rate: float = 0.01,
        learning_rate_decay: float = 0.9,
        central_nodes_embedding_path: Optional[str] = None,
        contextual_nodes_embedding_path: Optional[str] = None,
        normalize_by_degree: bool = False,
        stochastic_downsample_by_degree: Optional[bool] = False,
        normalize_learning_rate_by_degree: Optional[bool] = False,
        use_scale_free_distribution: Optional[bool] = True,
        random_state: int = 42,
        dtype: str = "f32",
        ring_bell: bool = False,
        enable_cache: bool = False,
        verbose: bool = True,
    ):
        super().__init__(**_forward(locals()))

    @classmethod
    def model_name(cls) -> str:
        return "Node2Vec SkipGram"


class Node2VecCBOWEnsmallen(Node2VecEnsmallen):
    """Node2Vec CBOW on the MI355X engine (reference: node2vec_cbow.py:6-166)."""

    def __init__(
        self,
        embedding_size: int = 100,
        epochs: int = 30,
        clipping_value: float = 6.0,
        number_of_negative_samples: int = 10,
        walk_length: int = 128,
        iterations: int = 10,
        window_size: int = 5,
        return_weight: float = 0.25,
        explore_weight: float = 4.0,
        change_node_type_weight: float = 1.0,
        change_edge_type_weight: float = 1.0,
        max_neighbours: Optional[int] = 100,
        learning_rate: float = 0.01,
        learning_rate_decay: float = 0.9,
        central_nodes_embedding_path: Optional[str] = None,
        contextual_nodes_embedding_path: Optional[str] = None,
        normalize_by_degree: bool = False,
        stochastic_downsample_by_degree: Optional[bool] = False,
        normalize_learning_rate_by_degree: Optional[bool] = False,
        use_scale_free_distribution: Optional[bool] = True,
        random_state: int = 42,
        dtype: str = "f32",
        ring_bell: bool = False,
        enable_cache: bool = False,
        verbose: bool = True,
    ):
        super().__init__(**_forward(locals()))

    @classmethod
    def model_name(cls) -> str:
        return "Node2Vec CBOW"


class Node2VecGloVeEnsmallen(Node2VecEnsmallen):
    """Node2Vec GloVe on the MI355X engine (reference: node2vec_glove.py:5-153): one walk of 512
    nodes per source node, co-occurrence counts inside the window, 100 epochs of GloVe SGD."""

    _REMOVED_PARAMETERS = ("change_node_type_weight", "change_edge_type_weight",
                           "number_of_negative_samples", "iterations")

    def __init__(
        self,
        embedding_size: int = 100,
        alpha: float = 0.75,
        epochs: int = 100,
        walk_length: int = 512,
        window_size: int = 5,
        return_weight: float = 0.25,
        explore_weight: float = 4.0,
        change_node_type_weight: float = 1.0,
        change_edge_type_weight: float = 1.0,
        max_neighbours: Optional[int] = 100,
        learning_rate: float = 0.05,
        learning_rate_decay: float = 0.9,
        central_nodes_embedding_path: Optional[str] = None,
        contextual_nodes_embedding_path: Optional[str] = None,
        normalize_by_degree: bool = False,
        dtype: str = "f32",
        random_state: int = 42,
        ring_bell: bool = False,
        enable_cache: bool = False,
        verbose: bool = True,
    ):
        super().__init__(iterations=1, **_forward(locals()))

    @classmethod
    def model_name(cls) -> str:
        return "Node2Vec GloVe"


class DeepWalkGloVeEnsmallen(Node2VecEnsmallen):
    """DeepWalk GloVe on the MI355X engine (reference: deepwalk_glove.py:5-123)."""

    _REMOVED_PARAMETERS = ("return_weight", "explore_weight", "change_node_type_weight",
                           "change_edge_type_weight", "number_of_negative_samples", "iterations")

    def __init__(
        self,
        embedding_size: int = 100,
        alpha: float = 0.75,
        epochs: int = 100,
        walk_length: int = 512,
        window_size: int = 5,
        max_neighbours: Optional[int] = 100,
        learning_rate: float = 0.05,
        learning_rate_decay: float = 0.99,
        central_nodes_embedding_path: Optional[str] = None,
        contextual_nodes_embedding_path: Optional[str] = None,
        normalize_by_degree: bool = False,
        dtype: str = "f32",
        random_state: int = 42,
        ring_bell: bool = False,
        enable_cache: bool = False,
        verbose: bool = True,
    ):
        # the reference leaves return / explore weight to the engine's defaults (first order)
        super().__init__(iterations=1, return_weight=1.0, explore_weight=1.0,
                         **_forward(locals()))

    @classmethod
    def model_name(cls) -> str:
        return "DeepWalk GloVe"


class _DeepWalkMixin:
    """DeepWalk = first-order walks: no return/explore weights (deepwalk_skipgram.py:9-32)."""

    _REMOVED_PARAMETERS = Node2VecEnsmallen._REMOVED_PARAMETERS + (
        "return_weight", "explore_weight",
    )


class DeepWalkSkipGramEnsmallen(_DeepWalkMixin, Node2VecEnsmallen):
    """DeepWalk SkipGram on the MI355X engine (reference: deepwalk_skipgram.py)."""

    def __init__(
        self,
        embedding_size: int = 100,
        epochs: int = 30,
        clipping_value: float = 6.0,
        number_of_negative_samples: int = 10,
        walk_length: int = 128,
        iterations: int = 10,
        window_size: int = 5,
        max_neighbours: Optional[int] = 100,
        learning_rate: float = 0.01,
        learning_rate_decay: float = 0.9,
        central_nodes_embedding_path: Optional[str] = None,
        contextual_nodes_embedding_path: Optional[str] = None,
        normalize_by_degree: bool = False,
        stochastic_downsample_by_degree: Optional[bool] = False,
        normalize_learning_rate_by_degree: Optional[bool] = False,
        use_scale_free_distribution: Optional[bool] = True,
        random_state: int = 42,
        dtype: str = "f32",
        ring_bell: bool = False,
        enable_cache: bool = False,
        verbose: bool = True,
    ):
        super().__init__(return_weight=1.0, explore_weight=1.0, **_forward(locals()))

    @classmethod
    def model_name(cls) -> str:
        return "DeepWalk SkipGram"


class DeepWalkCBOWEnsmallen(_DeepWalkMixin, Node2VecEnsmallen):
    """DeepWalk CBOW on the MI355X engine (reference: deepwalk_cbow.py)."""

    def __init__(
        self,
        embedding_size: int = 100,
        epochs: int = 30,
        clipping_value: float = 6.0,
        number_of_negative_samples: int = 10,
        walk_length: int = 128,
        iterations: int = 10,
        window_size: int = 5,
        max_neighbours: Optional[int] = 100,
        learning_rate: float = 0.01,
        learning_rate_decay: float = 0.9,
        central_nodes_embedding_path: Optional[str] = None,
        contextual_nodes_embedding_path: Optional[str] = None,
        normalize_by_degree: bool = False,
        stochastic_downsample_by_degree: Optional[bool] = False,
        normalize_learning_rate_by_degree: Optional[bool] = False,
        use_scale_free_distribution: Optional[bool] = True,
        random_state: int = 42,
        dtype: str = "f32",
        ring_bell: bool = False,
        enable_cache: bool = False,
        verbose: bool = True,
    ):
        super().__init__(return_weight=1.0, explore_weight=1.0, **_forward(locals()))

    @classmethod
    def model_name(cls) -> str:
        return "DeepWalk CBOW"


@abstract_class
class WalkletsEnsmallen(Node2VecEnsmallen):
    """Walklets on the MI355X engine (reference: walklets.py:7-150): one table pair per window
    scale, each `embedding_size // window_size` wide (:113); ``parameters()`` reports the total
    size again (:138-142)."""

    _REMOVED_PARAMETERS = ("alpha",)

    def __init__(
        self,
        embedding_size: int = 100,
        epochs: int = 10,
        clipping_value: float = 6.0,
        number_of_negative_samples: int = 10,
        walk_length: int = 128,
        iterations: int = 10,
        window_size: int = 4,
        return_weight: float = 1.0,
        explore_weight: float = 1.0,
        max_neighbours: Optional[int] = 100,
        learning_rate: float = 0.01,
        learning_rate_decay: float = 0.9,
        central_nodes_embedding_path: Optional[str] = None,
        contextual_nodes_embedding_path: Optional[str] = None,
        alpha: float = 0.75,
        normalize_by_degree: bool = False,
        stochastic_downsample_by_degree: Optional[bool] = False,
        normalize_learning_rate_by_degree: Optional[bool] = False,
        use_scale_free_distribution: Optional[bool] = True,
        random_state: int = 42,
        dtype: str = "f32",
        ring_bell: bool = False,
        enable_cache: bool = False,
    ):
        kwargs = _forward(locals())
        kwargs["embedding_size"] = embedding_size // window_size
        super().__init__(**kwargs)

    def parameters(self) -> Dict[str, Any]:
        parameters = super().parameters()
        parameters["embedding_size"] = parameters["embedding_size"] * parameters["window_size"]
        return parameters


class WalkletsSkipGramEnsmallen(WalkletsEnsmallen):
    """Walklets SkipGram on the MI355X engine (reference: walklets_skipgram.py:6-149)."""

    def __init__(
        self,
        embedding_size: int = 100,
        epochs: int = 30,
        clipping_value: float = 6.0,
        number_of_negative_samples: int = 10,
        walk_length: int = 128,
        iterations: int = 10,
        window_size: int = 4,
        return_weight: float = 1.0,
        explore_weight: float = 1.0,
        max_neighbours: Optional[int] = 100,
        learning_rate: float = 0.01,
        learning_rate_decay: float = 0.9,
        central_nodes_embedding_path: Optional[str] = None,
        contextual_nodes_embedding_path: Optional[str] = None,
        normalize_by_degree: bool = False,
        stochastic_downsample_by_degree: Optional[bool] = False,
        normalize_learning_rate_by_degree: Optional[bool] = False,
        use_scale_free_distribution: Optional[bool] = True,
        random_state: int = 42,
        dtype: str = "f32",
        ring_bell: bool = False,
        enable_cache: bool = False,
    ):
        super().__init__(**_forward(locals()))

    @classmethod
    def model_name(cls) -> str:
        return "Walklets SkipGram"


class WalkletsCBOWEnsmallen(WalkletsEnsmallen):
    """Walklets CBOW on the MI355X engine (reference: walklets_cbow.py:6-149)."""

    def __init__(
        self,
        embedding_size: int = 100,
        epochs: int = 30,
        clipping_value: float = 6.0,
        number_of_negative_samples: int = 10,
        walk_length: int = 128,
        iterations: int = 10,
        window_size: int = 4,
        return_weight: float = 1.0,
        explore_weight: float = 1.0,
        max_neighbours: Optional[int] = 100,
        learning_rate: float = 0.01,
        learning_rate_decay: float = 0.9,
        central_nodes_embedding_path: Optional[str] = None,
        contextual_nodes_embedding_path: Optional[str] = None,
        normalize_by_degree: bool = False,
        stochastic_downsample_by_degree: Optional[bool] = False,
        normalize_learning_rate_by_degree: Optional[bool] = False,
        use_scale_free_distribution: Optional[bool] = True,
        random_state: int = 42,
        dtype: str = "f32",
        ring_bell: bool = False,
        enable_cache: bool = False,
    ):
        super().__init__(**_forward(locals()))

    @classmethod
    def model_name(cls) -> str:
        return "Walklets CBOW"


class WalkletsGloVeEnsmallen(WalkletsEnsmallen):
    """Walklets GloVe on the MI355X engine (reference: walklets_glove.py:6-143): one GloVe table
    pair per window scale, on the co-occurrences exactly that many steps apart."""

    _REMOVED_PARAMETERS = ("number_of_negative_samples", "clipping_value", "iterations")

    def __init__(
        self,
        embedding_size: int = 100,
        epochs: int = 100,
        walk_length: int = 512,
        window_size: int = 4,
        return_weight: float = 1.0,
        explore_weight: float = 1.0,
        max_neighbours: Optional[int] = 100,
        learning_rate: float = 0.05,
        learning_rate_decay: float = 0.9,
        central_nodes_embedding_path: Optional[str] = None,
        contextual_nodes_embedding_path: Optional[str] = None,
        alpha: float = 0.75,
        normalize_by_degree: bool = False,
        stochastic_downsample_by_degree: Optional[bool] = False,
        normalize_learning_rate_by_degree: Optional[bool] = False,
        use_scale_free_distribution: Optional[bool] = True,
        random_state: int = 42,
        dtype: str = "f32",
        ring_bell: bool = False,
        enable_cache: bool = False,
    ):
        super().__init__(iterations=1, **_forward(locals()))

    @classmethod
    def model_name(cls) -> str:
        return "Walklets GloVe"


for _model in (Node2VecGloVeEnsmallen, DeepWalkGloVeEnsmallen, WalkletsGloVeEnsmallen):
    AbstractEmbeddingModel.register(_model)

for _model in (Node2VecSkipGramEnsmallen, Node2VecCBOWEnsmallen, DeepWalkSkipGramEnsmallen,
               DeepWalkCBOWEnsmallen, WalkletsSkipGramEnsmallen, WalkletsCBOWEnsmallen):
    AbstractEmbeddingModel.register(_model)

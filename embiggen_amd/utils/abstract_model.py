"""Model registry and base classes the embedders sit behind.

Minimal restatement of the reference's plug-in API for the Node Embedding task:
``AbstractModel`` (embiggen/utils/abstract_models/abstract_model.py:27-760) and
``AbstractEmbeddingModel`` (abstract_embedding_model.py:12-259) -- same method names, argument
meaning and error behaviour for everything the Node2Vec path touches: random-state validation
(:41-56), ``parameters`` / ``into_smoke_test`` (:146-154), ``consistent_hash`` (:555-564),
``set_random_state`` (:582-589), the model library (``register`` :721-749,
``get_model_from_library`` :626-700 incl. the "prefer Ensmallen" rule :670-675,
``find_available_models`` :702-719) and the graph validation / DataFrame guard of
``fit_transform`` (abstract_embedding_model.py:114-180, :236-247).
"""
import gzip
import hashlib
import json
import os
import pickle
import warnings
from typing import Any, Dict, List, Optional, Type

import pandas as pd

from .embedding_result import EmbeddingResult


def must_be_in_set(value, valid, what: str):
    """``userinput.utils.must_be_in_set`` stand-in: ValueError unless ``value`` is in ``valid``."""
    valid = list(valid)
    if value not in valid:
        raise ValueError(
            f"The provided {what} `{value}` is not valid. The supported values are: {valid}."
        )
    return value


def abstract_class(klass):
    """Marks a class as abstract for the registry (never registered itself)."""
    klass._gn2v_abstract = klass.__name__
    return klass


def _not_implemented(cls, method: str):
    return NotImplementedError(
        f"The `{method}` method must be implemented in the child classes of abstract model. "
        f"It was not implemented in the class {cls.__name__}."
    )


@abstract_class
class AbstractModel:
    MODELS_LIBRARY: Dict[str, Dict[str, Dict[str, Type["AbstractModel"]]]] = {}

    def __init__(self, random_state: Optional[int] = None):
        if self.is_stocastic() and random_state is None:
            raise ValueError(
                f"The provided model is stocastic, yet no random state was provided. Please do "
                f"provide a random state to the model {self.model_name()} from library "
                f"{self.library_name()} and task {self.task_name()}."
            )
        if not self.is_stocastic() and random_state is not None:
            raise ValueError(
                f"The provided model is not stocastic, yet a random state of `{random_state}` "
                f"was provided to the model {self.model_name()} from library "
                f"{self.library_name()} and task {self.task_name()}."
            )
        self._random_state = random_state

    # ------------------------------------------------------------------ parameters
    @classmethod
    def smoke_test_parameters(cls) -> Dict[str, Any]:
        raise _not_implemented(cls, "smoke_test_parameters")

    def parameters(self) -> Dict[str, Any]:
        return {} if self._random_state is None else {"random_state": self._random_state}

    def into_smoke_test(self):
        return self.__class__(**{**self.parameters(), **self.smoke_test_parameters()})

    def clone(self):
        return self.__class__(**self.parameters())

    def consistent_hash(self) -> str:
        payload = dict(
            **self.parameters(), model_name=self.model_name(),
            library_name=self.library_name(), task_name=self.task_name(),
        )
        return hashlib.sha256(
            json.dumps(payload, sort_keys=True, default=str).encode()
        ).hexdigest()

    def set_random_state(self, random_state: int):
        if not self.is_stocastic():
            raise ValueError(
                "It does not make sense to set the random state of a non-stocastic model."
            )
        self._random_state = random_state

    # ------------------------------------------------------------------ capabilities
    @classmethod
    def task_name(cls) -> str:
        raise _not_implemented(cls, "task_name")

    @classmethod
    def library_name(cls) -> str:
        raise _not_implemented(cls, "library_name")

    @classmethod
    def model_name(cls) -> str:
        raise _not_implemented(cls, "model_name")

    @classmethod
    def is_stocastic(cls) -> bool:
        raise _not_implemented(cls, "is_stocastic")

    @classmethod
    def is_topological(cls) -> bool:
        raise _not_implemented(cls, "is_topological")

    @staticmethod
    def is_available() -> bool:
        return True

    @classmethod
    def requires_edge_weights(cls) -> bool:
        raise _not_implemented(cls, "requires_edge_weights")

    @classmethod
    def requires_positive_edge_weights(cls) -> bool:
        raise _not_implemented(cls, "requires_positive_edge_weights")

    @classmethod
    def can_use_edge_weights(cls) -> bool:
        raise _not_implemented(cls, "can_use_edge_weights")

    @classmethod
    def requires_node_types(cls) -> bool:
        raise _not_implemented(cls, "requires_node_types")

    @classmethod
    def can_use_node_types(cls) -> bool:
        raise _not_implemented(cls, "can_use_node_types")

    @classmethod
    def requires_edge_types(cls) -> bool:
        raise _not_implemented(cls, "requires_edge_types")

    @classmethod
    def can_use_edge_types(cls) -> bool:
        raise _not_implemented(cls, "can_use_edge_types")

    @classmethod
    def requires_edge_type_features(cls) -> bool:
        return False

    @classmethod
    def can_use_edge_type_features(cls) -> bool:
        return False

    @classmethod
    def requires_edge_features(cls) -> bool:
        return False

    @classmethod
    def can_use_edge_features(cls) -> bool:
        return False

    # ------------------------------------------------------------------ model library
    @staticmethod
    def register(model_class: Type["AbstractModel"]):
        task = AbstractModel.MODELS_LIBRARY.setdefault(model_class.task_name(), {})
        libraries = task.setdefault(model_class.model_name(), {})
        libraries.setdefault(model_class.library_name(), model_class)
        return model_class

    @staticmethod
    def get_task_data(model_name: str, task_name: str) -> Dict[str, Type["AbstractModel"]]:
        if len(model_name) == 0:
            raise ValueError("The provided model name is empty.")
        if len(task_name) == 0:
            raise ValueError("The provided task name is empty.")
        task_name = must_be_in_set(task_name, AbstractModel.MODELS_LIBRARY, "task name")
        model_name = must_be_in_set(
            model_name, AbstractModel.MODELS_LIBRARY[task_name], "model name"
        )
        return AbstractModel.MODELS_LIBRARY[task_name][model_name]

    @staticmethod
    def get_library_data(model_name: str, task_name: str, library_name: str):
        task_data = AbstractModel.get_task_data(model_name, task_name)
        if len(library_name) == 0:
            raise ValueError("The provided library name is empty.")
        return task_data[must_be_in_set(library_name, task_data.keys(), "library name")]

    @classmethod
    def get_model_from_library(cls, model_name: str, task_name: Optional[str] = None,
                               library_name: Optional[str] = None) -> Type["AbstractModel"]:
        if task_name is None:
            try:
                task_name = cls.task_name()
            except NotImplementedError as exception:
                frame = get_models_dataframe()
                if len(frame) == 0 or model_name not in frame.model_name.values:
                    raise ValueError(
                        f"The requested model `{model_name}` is not available. Please do "
                        "provide a valid model name to resolve this ambiguity."
                    ) from exception
                task_name = frame[frame.model_name == model_name].iloc[0].task_name
        task_data = AbstractModel.get_task_data(model_name, task_name)
        if library_name is None:
            names = list(task_data.keys())
            if len(names) == 1:
                library_name = names[0]
            elif "Ensmallen" in names:
                library_name = "Ensmallen"
            else:
                raise ValueError(
                    f"The requested model `{model_name}` is available for multiple libraries "
                    f"({names}) and no specific library was requested."
                )
        model_class = AbstractModel.get_library_data(model_name, task_name, library_name)
        if not model_class.is_available():
            model_class()  # raises the stub's helpful ModuleNotFoundError
        return model_class

    @staticmethod
    def find_available_models(model_name: str, task_name: str) -> List[Type["AbstractModel"]]:
        return [
            model for model in AbstractModel.get_task_data(model_name, task_name).values()
            if model.is_available()
        ]


def get_model_metadata(model_class: Type[AbstractModel]) -> Dict[str, Any]:
    return {
        "model_name": model_class.model_name(),
        "task_name": model_class.task_name(),
        "library_name": model_class.library_name(),
        "available": model_class.is_available(),
        "requires_node_types": model_class.requires_node_types(),
        "can_use_node_types": model_class.requires_node_types() or model_class.can_use_node_types(),
        "requires_edge_types": model_class.requires_edge_types(),
        "can_use_edge_types": model_class.requires_edge_types() or model_class.can_use_edge_types(),
        "requires_edge_weights": model_class.requires_edge_weights(),
        "can_use_edge_weights": model_class.requires_edge_weights()
        or model_class.can_use_edge_weights(),
        "requires_positive_edge_weights": model_class.requires_positive_edge_weights(),
    }


def get_models_dataframe() -> pd.DataFrame:
    return pd.DataFrame([
        get_model_metadata(model_class)
        for models in AbstractModel.MODELS_LIBRARY.values()
        for libraries in models.values()
        for model_class in libraries.values()
    ])


def get_available_models_for_node_embedding() -> pd.DataFrame:
    frame = get_models_dataframe()
    return frame[(frame.task_name == "Node Embedding") & frame.available]


@abstract_class
class AbstractEmbeddingModel(AbstractModel):
    def __init__(self, embedding_size: Optional[int] = None, enable_cache: bool = False,
                 ring_bell: bool = False, random_state: Optional[int] = None):
        super().__init__(random_state=random_state)
        if (embedding_size is not None and not isinstance(embedding_size, int)
                or embedding_size == 0):
            raise ValueError(
                "The embedding size, if provided, should be a strictly positive integer "
                f"but {embedding_size} was provided."
            )
        self._embedding_size = embedding_size
        self._enable_cache = enable_cache
        self._ring_bell = ring_bell  # accepted for signature parity; no sound is played

    def parameters(self) -> Dict[str, Any]:
        extra = {} if self._embedding_size is None else {"embedding_size": self._embedding_size}
        return dict(**super().parameters(), **extra)

    @classmethod
    def task_name(cls) -> str:
        return "Node Embedding"

    @classmethod
    def requires_nodes_sorted_by_decreasing_node_degree(cls) -> bool:
        raise _not_implemented(cls, "requires_nodes_sorted_by_decreasing_node_degree")

    @classmethod
    def get_minimum_required_number_of_node_types(cls) -> int:
        return 0

    def _fit_transform(self, graph, return_dataframe: bool = True) -> EmbeddingResult:
        raise _not_implemented(type(self), "_fit_transform")

    # ------------------------------------------------------------------ cache
    def _cache_path(self, graph, return_dataframe: bool) -> str:
        # the reference's `@Cache {_hash}` also hashes the graph argument: the key depends on the
        # graph's content, not just on its name (abstract_embedding_model.py:91-95)
        digest = graph.content_digest() if hasattr(graph, "content_digest") else (
            f"{graph.get_number_of_nodes()}:{graph.get_number_of_directed_edges()}")
        key = hashlib.sha256(
            json.dumps({"model": self.consistent_hash(), "df": return_dataframe,
                        "graph": digest}).encode()
        ).hexdigest()
        return os.path.join("embedding", self.model_name(), self.library_name(),
                            graph.get_name(), f"{key}.pkl.gz")

    def _cached_fit_transform(self, graph, return_dataframe: bool = True) -> EmbeddingResult:
        if self._enable_cache:
            path = self._cache_path(graph, return_dataframe)
            if os.path.exists(path):
                with gzip.open(path, "rb") as handle:
                    return EmbeddingResult.load(pickle.load(handle))
        result = self._validated_fit_transform(graph, return_dataframe)
        if self._enable_cache:
            os.makedirs(os.path.dirname(path), exist_ok=True)
            with gzip.open(path, "wb") as handle:
                pickle.dump(result.dump(), handle)
        return result

    def _validated_fit_transform(self, graph, return_dataframe: bool) -> EmbeddingResult:
        name = graph.get_name()
        if not graph.has_nodes():
            raise ValueError(f"The provided graph {name} is empty.")
        if (self.requires_nodes_sorted_by_decreasing_node_degree()
                and not graph.has_nodes_sorted_by_decreasing_outbound_node_degree()):
            raise ValueError(
                f"The given graph {name} does not have the nodes sorted by decreasing degree; "
                "use `graph.sort_by_decreasing_outbound_node_degree()`."
            )
        if self.requires_node_types() and not graph.has_node_types():
            raise ValueError(
                f"The provided graph {name} does not have node types, but the "
                f"{self.model_name()} requires node types."
            )
        if self.requires_node_types() and graph.get_number_of_node_types() <= 1:
            raise ValueError(
                f"The {self.model_name()} requires the graph to have at least "
                f"{self.get_minimum_required_number_of_node_types()} node types, but the "
                f"provided one has {graph.get_number_of_node_types()} node types."
            )
        if self.requires_edge_types() and not graph.has_edge_types():
            raise ValueError(
                f"The provided graph {name} does not have edge types, but the "
                f"{self.model_name()} requires edge types."
            )
        if self.requires_edge_weights() and not graph.has_edge_weights():
            raise ValueError(
                f"The provided graph {name} does not have edge weights, but the "
                f"{self.model_name()} requires edge weights."
            )
        if (self.requires_positive_edge_weights() and graph.has_edge_weights()
                and graph.has_negative_edge_weights()):
            raise ValueError(
                f"The provided graph {name} has negative edge weights, but the "
                f"{self.model_name()} requires strictly positive edge weights."
            )
        if self.is_topological():
            if not graph.has_edges():
                raise ValueError(f"The provided graph {name} does not have edges.")
            if graph.has_disconnected_nodes():
                warnings.warn(
                    f"Please be advised that the {name} graph contains "
                    f"{graph.get_number_of_disconnected_nodes()} disconnected nodes. "
                    "Topological node embedding algorithms such as CBOW and SkipGram cannot "
                    "provide meaningful embeddings for these nodes."
                )
        result = self._fit_transform(graph=graph, return_dataframe=return_dataframe)
        if not isinstance(result, EmbeddingResult):
            raise NotImplementedError(
                f"The embedding result produced by the {self.model_name()} method from the "
                f"library {self.library_name()} implemented in the class "
                f"{self.__class__.__name__} is not an EmbeddingResult but a {type(result)}."
            )
        return result

    def fit_transform(self, graph, repository: Optional[str] = None,
                      version: Optional[str] = None,
                      return_dataframe: bool = True) -> EmbeddingResult:
        """Execute the embedding on the provided graph (a ``CSRGraph`` or any object exposing
        the same getters).  Graph names are resolved by the reference through
        ``ensmallen.datasets`` (network download, abstract_embedding_model.py:230-235); here only
        the built-in offline datasets are resolvable."""
        if isinstance(graph, str):
            from ..datasets import get_dataset

            graph = get_dataset(name=graph, repository=repository, version=version)
        if return_dataframe and graph.get_number_of_nodes() > 100_000_000:
            raise ValueError(
                "We cowardly refuse to execute this embedding with the added requirement to "
                f"also return the dataframe version: the graph has {graph.get_number_of_nodes()} "
                "nodes and creating a DataFrame would most likely cause an OOM."
            )
        return self._cached_fit_transform(graph=graph, return_dataframe=return_dataframe)

"""Result container returned by every embedding model.

Behavioural twin of the reference's ``EmbeddingResult``
(embiggen/utils/abstract_models/embedding_result.py:11-334): same constructor arguments, same
validation (type / empty / NaN / Inf -> ``ValueError``, all-zero -> warning, checks skipped above
1 000 000 rows, :78-79), same getters and error conventions (tests/test_embedding_result.py:12-90),
``dump``/``load`` round trip (:321-334) and method proxying when it wraps a single embedding
(:114-129).  Written from that behaviour, not from its text.
"""
import inspect
import types
import warnings
from typing import Dict, List, Optional, Union

import numpy as np
import pandas as pd

Embedding = Union[np.ndarray, pd.DataFrame]
_KINDS = (
    ("node_embeddings", "node embedding", "node"),
    ("edge_embeddings", "edge embedding", "edge"),
    ("node_type_embeddings", "node type embedding", "node type"),
    ("edge_type_embeddings", "edge type embedding", "edge type"),
)
_MAX_CHECKED_ROWS = 1_000_000


def _as_list(value):
    if value is None or isinstance(value, list):
        return value
    return [value]


class EmbeddingResult:
    def __init__(
        self,
        embedding_method_name: str,
        node_embeddings: Optional[Union[Embedding, List[Embedding]]] = None,
        edge_embeddings: Optional[Union[Embedding, List[Embedding]]] = None,
        node_type_embeddings: Optional[Union[Embedding, List[Embedding]]] = None,
        edge_type_embeddings: Optional[Union[Embedding, List[Embedding]]] = None,
    ):
        given = dict(
            node_embeddings=_as_list(node_embeddings),
            edge_embeddings=_as_list(edge_embeddings),
            node_type_embeddings=_as_list(node_type_embeddings),
            edge_type_embeddings=_as_list(edge_type_embeddings),
        )
        for attribute, label, _ in _KINDS:
            for embedding in given[attribute] or ():
                self._validate(embedding, label, embedding_method_name)

        self._embedding_method_name = embedding_method_name
        self._node_embeddings = given["node_embeddings"]
        self._edge_embeddings = given["edge_embeddings"]
        self._node_type_embeddings = given["node_type_embeddings"]
        self._edge_type_embeddings = given["edge_type_embeddings"]

        if self.is_single_embedding():
            self._proxy_methods_of(self.get_single_embedding())

    @staticmethod
    def _validate(embedding, label: str, method: str):
        if not isinstance(embedding, (np.ndarray, pd.DataFrame)):
            raise ValueError(
                f"A {label} computed with the {method} method is neither a numpy array nor a "
                f"pandas DataFrame but a `{type(embedding)}` object."
            )
        if embedding.shape[0] == 0:
            raise ValueError(f"A {label} computed with the {method} method is empty.")
        if embedding.shape[0] > _MAX_CHECKED_ROWS:
            return
        values = embedding.to_numpy() if isinstance(embedding, pd.DataFrame) else embedding
        if np.isnan(values).any():
            raise ValueError(f"A {label} computed with the {method} method contains NaN values.")
        infinite = np.isinf(values)
        if infinite.any():
            raise ValueError(
                f"A {label} computed with the {method} method contains "
                f"{int(infinite.sum())} infinite values."
            )
        if np.isclose(values, 0.0).all():
            warnings.warn(
                f"A {label} computed with the {method} method contains exclusively zeros."
            )

    def _proxy_methods_of(self, embedding):
        """Expose the bound methods of the single wrapped embedding on the result itself."""
        for method_name, method in inspect.getmembers(
            embedding, lambda member: isinstance(member, types.MethodType)
        ):
            def forward(*args, _name=method_name, **kwargs):
                return getattr(embedding, _name)(*args, **kwargs)

            forward.__doc__ = method.__doc__
            forward.__name__ = method.__name__
            setattr(self, method_name, forward)

    # ------------------------------------------------------------------ bookkeeping
    def _lists(self):
        return (self._node_embeddings, self._edge_embeddings, self._node_type_embeddings,
                self._edge_type_embeddings)

    def number_of_embeddings(self) -> int:
        return sum(len(lst) for lst in self._lists() if lst is not None)

    def is_single_embedding(self) -> bool:
        return self.number_of_embeddings() == 1

    def get_single_embedding(self) -> Embedding:
        assert self.is_single_embedding()
        for lst in self._lists():
            if lst is not None:
                return lst[0]

    @property
    def embedding_method_name(self) -> str:
        return self._embedding_method_name

    # ------------------------------------------------------------------ getters
    def _all(self, attribute: str, what: str) -> List[Embedding]:
        lst = getattr(self, attribute)
        if lst is None:
            raise ValueError(
                f"The {what} embedding were requested but they were not computed by the "
                f"{self._embedding_method_name} method."
            )
        return lst

    def _at(self, attribute: str, what: str, index: int) -> Embedding:
        lst = self._all(attribute, what)
        if index >= len(lst):
            raise ValueError(
                f"The {what} embedding computed with the {self._embedding_method_name} method "
                f"are {len(lst)}, but you requested the embedding in position {index}."
            )
        return lst[index]

    def get_all_node_embedding(self) -> List[Embedding]:
        """All node embeddings; SkipGram-style models return two (input-side table first)."""
        return self._all("_node_embeddings", "node")

    def get_all_edge_embedding(self) -> List[Embedding]:
        return self._all("_edge_embeddings", "edge")

    def get_all_node_type_embeddings(self) -> List[Embedding]:
        return self._all("_node_type_embeddings", "node types")

    def get_all_edge_type_embeddings(self) -> List[Embedding]:
        return self._all("_edge_type_embeddings", "edge types")

    def get_node_embedding_from_index(self, index: int) -> Embedding:
        return self._at("_node_embeddings", "node", index)

    def get_edge_embedding_from_index(self, index: int) -> Embedding:
        return self._at("_edge_embeddings", "edge", index)

    def get_node_type_embedding_from_index(self, index: int) -> Embedding:
        return self._at("_node_type_embeddings", "node type", index)

    def get_edge_type_embedding_from_index(self, index: int) -> Embedding:
        return self._at("_edge_type_embeddings", "edge type", index)

    # ------------------------------------------------------------------ cache round trip
    def dump(self) -> Dict[str, object]:
        return {
            "embedding_method_name": self._embedding_method_name,
            "node_embeddings": self._node_embeddings,
            "edge_embeddings": self._edge_embeddings,
            "node_type_embeddings": self._node_type_embeddings,
            "edge_type_embeddings": self._edge_type_embeddings,
        }

    @staticmethod
    def load(cached: Dict[str, object]) -> "EmbeddingResult":
        return EmbeddingResult(**cached)

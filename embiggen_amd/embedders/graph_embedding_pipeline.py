"""``embed_graph``: the convenience entry point most callers use.

Same signature and error conventions as the reference
(embiggen/embedders/graph_embedding_pipeline.py:10-107): a model name is looked up in the model
library (:59-63), kwargs together with a model instance raise ``ValueError`` (:64-70), a
non-embedding-model raises ``ValueError`` (:72-76), smoke-test conversion failures and any
exception of ``fit_transform`` are re-raised as ``ValueError`` (:78-89, :94-107).
"""
from typing import Dict, Optional, Type, Union

from ..utils.abstract_model import AbstractEmbeddingModel
from ..utils.embedding_result import EmbeddingResult


def embed_graph(
    graph,
    embedding_model: Union[str, Type[AbstractEmbeddingModel]],
    repository: Optional[str] = None,
    version: Optional[str] = None,
    library_name: Optional[str] = None,
    smoke_test: bool = False,
    return_dataframe: bool = True,
    **kwargs: Dict
) -> EmbeddingResult:
    if isinstance(graph, str):
        from ..datasets import get_dataset

        graph = get_dataset(graph, repository=repository, version=version)

    if isinstance(embedding_model, str):
        embedding_model = AbstractEmbeddingModel.get_model_from_library(
            model_name=embedding_model, library_name=library_name,
        )(**kwargs)
    elif kwargs:
        raise ValueError(
            "You have provided both an embedding model instance and kwargs that would normally "
            "be forwarded to the creation of the model: it is unclear what to do with them."
        )

    if not issubclass(embedding_model.__class__, AbstractEmbeddingModel):
        raise ValueError(
            "The provided object is not an embedding model, that is, it does not extend the "
            "class `AbstractEmbeddingModel`."
        )

    if smoke_test:
        try:
            embedding_model = embedding_model.into_smoke_test()
        except Exception as e:
            raise ValueError(
                "An exception was raised while creating a smoke test version of the model "
                f"{embedding_model.model_name()} from the library {library_name}, class "
                f"{embedding_model.__class__.__name__}: {e}"
            ) from e

    if embedding_model.requires_nodes_sorted_by_decreasing_node_degree():
        graph = graph.sort_by_decreasing_outbound_node_degree()

    try:
        return embedding_model.fit_transform(graph, return_dataframe=return_dataframe)
    except Exception as e:
        raise ValueError(
            f"An exception was raised while computing a node embedding on the graph "
            f"{graph.get_name()} using the model {embedding_model.model_name()} from the library "
            f"{library_name}, class {embedding_model.__class__.__name__}: {e}"
        ) from e

"""Keyword-argument normalisation for the walk-based embedders.

Same contract as the reference's ``normalize_kwargs`` (embiggen/utils/normalize_kwargs.py:74-136):
each known key is coerced to its declared type (values already of a valid type are left alone),
a value that cannot be coerced raises ``TypeError`` (:117-125) and any key outside the schema
raises ``NotImplementedError`` (:127-134).  The schema below lists only the keys of the Node2Vec /
DeepWalk / Walklets constructors, with the types the reference declares for them in
``normalization_schemas.json`` (checked against it by tests/golden/api_defaults.json).
"""
from typing import Any, Dict, Tuple

_TYPES = {"bool": bool, "str": str, "None": type(None), "int": int, "float": float}

# key -> accepted type names, in coercion order
SCHEMA: Dict[str, Tuple[str, ...]] = {
    "embedding_size": ("int",),
    "epochs": ("int",),
    "clipping_value": ("float",),
    "number_of_negative_samples": ("int",),
    "walk_length": ("int",),
    "iterations": ("int",),
    "window_size": ("int",),
    "return_weight": ("float",),
    "explore_weight": ("float",),
    "change_node_type_weight": ("float",),
    "change_edge_type_weight": ("float",),
    "max_neighbours": ("int",),
    "learning_rate": ("float", "str"),
    "learning_rate_decay": ("float",),
    "central_nodes_embedding_path": ("str", "None"),
    "contextual_nodes_embedding_path": ("str", "None"),
    "normalize_by_degree": ("bool",),
    "stochastic_downsample_by_degree": ("bool",),
    "normalize_learning_rate_by_degree": ("bool",),
    "use_scale_free_distribution": ("bool",),
    "random_state": ("int",),
    "dtype": ("str",),
    "verbose": ("bool",),
    "alpha": ("float",),
    # engine-specific extensions (not in the reference)
    "deterministic": ("bool",),
    "update_mode": ("str",),
    "device": ("int",),
}


def _coerce(type_names: Tuple[str, ...], value: Any) -> Any:
    for name in type_names:
        if name == "bool":
            return bool(value)
        if name == "int":
            try:
                return int(value)
            except ValueError:
                continue
        if name == "float":
            return float(value)
        if name == "str":
            return str(value)
    raise NotImplementedError(f"No coercion available for {type_names} and value {value!r}.")


def normalize_kwargs(model, kwargs: Dict[str, Any]) -> Dict[str, Any]:
    """Coerce ``kwargs`` in place to the schema types and return it."""
    unsupported = [key for key in kwargs if key not in SCHEMA]
    for key, value in kwargs.items():
        if key not in SCHEMA:
            continue
        names = SCHEMA[key]
        if isinstance(value, tuple(_TYPES[n] for n in names)):
            continue
        try:
            kwargs[key] = _coerce(names, value)
        except (TypeError, ValueError) as exception:
            raise TypeError(
                f"Parameter {key} has value \"{value}\" of type {type(value)} but the expected "
                f"type is {list(names)}. The model is {model.model_name()} from library "
                f"{model.library_name()} for the task {model.task_name()}."
            ) from exception
    if unsupported:
        raise NotImplementedError(
            f"The following parameters are not supported: {unsupported}. The model is "
            f"{model.model_name()} from library {model.library_name()} for the task "
            f"{model.task_name()}."
        )
    return kwargs

"""Offline dataset resolution.

The reference resolves graph names through ``ensmallen.datasets.get_dataset`` which downloads
from the network (embiggen/utils/abstract_models/abstract_embedding_model.py:230-235,
embiggen/utils/pipeline.py:97-102).  Neither this container nor the GPU box has network access,
so only the built-in fixtures / generators are resolvable by name.
"""
from typing import Optional

from .graph import CSRGraph, barabasi_albert, karate_club

_BUILTIN = {
    "karateclub": karate_club,
    "karate": karate_club,
    "zachary": karate_club,
}


def get_dataset(name: str, repository: Optional[str] = None,
                version: Optional[str] = None) -> CSRGraph:
    """Return a built-in graph by name; ``BA:<nodes>:<m>[:<seed>]`` generates a seeded
    Barabasi-Albert graph on the GPU."""
    key = name.replace(" ", "").replace("_", "").lower()
    if key in _BUILTIN:
        return _BUILTIN[key]()
    if key.startswith("ba:"):
        parts = key.split(":")
        seed = int(parts[3]) if len(parts) > 3 else 42
        return barabasi_albert(int(parts[1]), int(parts[2]), seed)
    raise ValueError(
        f"The graph `{name}` (repository {repository}, version {version}) cannot be retrieved: "
        "automatic graph retrieval needs ensmallen and network access. Available offline: "
        f"{sorted(set(_BUILTIN))} and `BA:<nodes>:<m>[:<seed>]`; or pass a CSRGraph."
    )

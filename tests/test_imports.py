"""The package's export lists resolve -- the check of the reference's
tests/test_all_imports_are_consistent.py:6-58 (every name of every ``__all__`` importable, no
duplicates, recursively over the package's own submodules) and of tests/test_version.py:10-12
(a PEP 440 version string), applied to embiggen_amd."""
import inspect
import re

import embiggen_amd


def _check(module, seen, problems):
    if module.__name__ in seen:
        return
    seen.add(module.__name__)
    names = getattr(module, "__all__", None)
    if names is not None:
        missing = sorted(n for n in names if not hasattr(module, n))
        duplicated = sorted({n for n in names if names.count(n) > 1})
        if missing:
            problems.append(f"{module.__name__}.__all__ names {missing} that cannot be imported")
        if duplicated:
            problems.append(f"{module.__name__}.__all__ repeats {duplicated}")
    for value in vars(module).values():
        owner = inspect.getmodule(value)
        if inspect.ismodule(value) and owner is not None and owner.__name__.startswith(
                "embiggen_amd"):
            _check(value, seen, problems)


def test_export_lists_resolve():
    seen, problems = set(), []
    _check(embiggen_amd, seen, problems)
    assert not problems, "\n".join(problems)
    assert {"embiggen_amd", "embiggen_amd.embedders", "embiggen_amd.utils",
            "embiggen_amd.sequences", "embiggen_amd.embedding_transformers"} <= seen


def test_every_model_of_the_path_is_exported_at_the_top_level():
    """What `from embiggen import ...` offers for this path (embedders/ensmallen_embedders/__init__.py,
    embedders/__init__.py, sequences, embedding_transformers) is importable from the package root."""
    for name in ("Node2VecSkipGramEnsmallen", "Node2VecCBOWEnsmallen", "Node2VecGloVeEnsmallen",
                 "DeepWalkSkipGramEnsmallen", "DeepWalkCBOWEnsmallen", "DeepWalkGloVeEnsmallen",
                 "WalkletsSkipGramEnsmallen", "WalkletsCBOWEnsmallen", "WalkletsGloVeEnsmallen",
                 "embed_graph", "get_available_models_for_node_embedding", "Node2VecSequence",
                 "EdgeTransformer", "NodeTransformer", "GraphTransformer", "EmbeddingResult"):
        assert name in embiggen_amd.__all__ and hasattr(embiggen_amd, name), name


def test_version_string():
    assert re.fullmatch(r"\d+(\.\d+)*((a|b|rc)\d+)?(\.post\d+)?(\.dev\d+)?",
                        embiggen_amd.__version__)

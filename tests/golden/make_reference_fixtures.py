"""Generates the reference-derived fixtures in this directory.  Run in the build container only
(needs /root/reference; the GPU box never runs this):

    python tests/golden/make_reference_fixtures.py

1. ``embedding_result_cases.json`` -- outcomes of the reference's own ``EmbeddingResult``
   (embiggen/utils/abstract_models/embedding_result.py, the only reference module importable
   without ensmallen) on the scenarios of the reference's tests/test_embedding_result.py:12-90
   plus a few more; our ``EmbeddingResult`` must reproduce every outcome.
2. ``api_defaults.json`` -- constructor signatures/defaults of the walk-based wrappers,
   ``smoke_test_parameters`` and the "removed" parameter lists, read from the reference sources
   with ``ast`` (no import), and the declared types of those kwargs from
   ``embiggen/utils/normalization_schemas.json``.

3. ``edge_embedding_cases.npz`` -- outputs of the reference's 12 edge operators on seeded inputs.
4. ``transformer_cases.npz`` / ``transformer_cases.json`` -- outcomes (arrays / exception type
   names) of the reference's NodeTransformer, EdgeTransformer and GraphTransformer on the scenario
   table ``helpers.transformer_cases``; the graph argument is a small pure-Python stand-in for
   ``ensmallen.Graph`` defined below (the reference only calls getters on it).
5. ``prediction_cases.npz`` / ``prediction_cases.json`` -- the same for its EdgePrediction-,
   EdgeLabelPrediction- and NodeLabelPredictionTransformer on ``helpers.prediction_cases``
   (the (X, y) pair of every scenario as one array), over the stand-in graphs of
   ``helpers.prediction_graph_specs``.

Only data (inputs -> expected outcomes, names -> default values) is written; no reference source
text is stored.
"""
import ast
import importlib.util
import json
import os
import sys


REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from helpers import (prediction_cases, probe, scenarios, transformer_cases,  # noqa: E402
                     transformer_graph_spec)


def embedding_result_cases():
    spec = importlib.util.spec_from_file_location(
        "ref_embedding_result",
        os.path.join(REF, "embiggen/utils/abstract_models/embedding_result.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return {name: probe(mod.EmbeddingResult, kw) for name, kw in scenarios().items()}


def _literal(node):
    try:
        return ast.literal_eval(node)
    except Exception:  # noqa: BLE001
        return ast.unparse(node)


def class_defaults(path, class_name):
    tree = ast.parse(open(path).read())
    out = {}
    for node in ast.walk(tree):
        if isinstance(node, ast.ClassDef) and node.name == class_name:
            for fn in node.body:
                if isinstance(fn, ast.FunctionDef) and fn.name == "__init__":
                    args = fn.args.args[1:]
                    defaults = fn.args.defaults
                    names = [a.arg for a in args]
                    out["init"] = {
                        n: _literal(d) for n, d in zip(names[len(names) - len(defaults):], defaults)
                    }
                    out["init_order"] = names
                if isinstance(fn, ast.FunctionDef) and fn.name == "model_name":
                    out["model_name"] = [
                        _literal(n.value) for n in ast.walk(fn) if isinstance(n, ast.Return)][0]
                if isinstance(fn, ast.FunctionDef) and fn.name == "parameters":
                    for n in ast.walk(fn):
                        if isinstance(n, ast.Assign) and getattr(n.targets[0], "id", "") == "removed":
                            out["removed"] = _literal(n.value)
                if isinstance(fn, ast.FunctionDef) and fn.name == "smoke_test_parameters":
                    for n in ast.walk(fn):
                        if isinstance(n, ast.Return) and isinstance(n.value, ast.Call):
                            out["smoke_test_parameters"] = {
                                kw.arg: _literal(kw.value) for kw in n.value.keywords}
                if isinstance(fn, ast.FunctionDef) and fn.name in (
                        "task_name", "library_name", "is_topological", "is_stocastic",
                        "requires_nodes_sorted_by_decreasing_node_degree", "requires_edge_weights",
                        "requires_positive_edge_weights", "can_use_edge_weights",
                        "can_use_node_types", "can_use_edge_types", "requires_node_types",
                        "requires_edge_types"):
                    rets = [n for n in ast.walk(fn) if isinstance(n, ast.Return)]
                    if len(rets) == 1:
                        out.setdefault("flags", {})[fn.name] = _literal(rets[0].value)
    return out


def api_defaults():
    base = os.path.join(REF, "embiggen/embedders/ensmallen_embedders")
    classes = {
        "Node2VecSkipGramEnsmallen": "node2vec_skipgram.py",
        "Node2VecCBOWEnsmallen": "node2vec_cbow.py",
        "DeepWalkSkipGramEnsmallen": "deepwalk_skipgram.py",
        "DeepWalkCBOWEnsmallen": "deepwalk_cbow.py",
        "WalkletsSkipGramEnsmallen": "walklets_skipgram.py",
        "WalkletsCBOWEnsmallen": "walklets_cbow.py",
        "WalkletsEnsmallen": "walklets.py",
        "Node2VecGloVeEnsmallen": "node2vec_glove.py",
        "DeepWalkGloVeEnsmallen": "deepwalk_glove.py",
        "WalkletsGloVeEnsmallen": "walklets_glove.py",
        "Node2VecEnsmallen": "node2vec.py",
        "EnsmallenEmbedder": "ensmallen_embedder.py",
    }
    out = {name: class_defaults(os.path.join(base, f), name) for name, f in classes.items()}
    schema = json.load(open(os.path.join(REF, "embiggen/utils/normalization_schemas.json")))
    keys = set()
    for name in classes:
        keys.update(out[name].get("init", {}).keys())
    keys.update(["embedding_size", "random_state", "alpha"])
    out["schema_types"] = {k: schema[k] for k in sorted(keys) if k in schema}
    out["sequence_defaults"] = class_defaults(
        os.path.join(REF, "embiggen/sequences/tensorflow_sequences/node2vec_sequence.py"),
        "Node2VecSequence").get("init", {})
    return out


def edge_embedding_cases():
    """Outputs of the reference's own edge operators on seeded inputs.  The module imports
    `ensmallen` and `userinput` at the top although the operators are plain numpy, so those two
    imports (and the sibling NodeTransformer import) are satisfied with empty placeholder modules
    for the duration of this call; no reference function that is executed touches them."""
    import types

    import numpy as np

    placeholders = {}
    for name in ("ensmallen", "ensmallen.express_measures", "userinput", "userinput.utils",
                 "embiggen", "embiggen.embedding_transformers",
                 "embiggen.embedding_transformers.node_transformer"):
        if name not in sys.modules:
            placeholders[name] = sys.modules[name] = types.ModuleType(name)
    sys.modules["ensmallen"].express_measures = sys.modules["ensmallen.express_measures"]
    sys.modules["userinput.utils"].must_be_in_set = lambda *a, **k: a[0]
    sys.modules["embiggen.embedding_transformers.node_transformer"].NodeTransformer = object
    try:
        spec = importlib.util.spec_from_file_location(
            "ref_edge_transformer",
            os.path.join(REF, "embiggen/embedding_transformers/edge_transformer.py"))
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        methods = dict(mod.EdgeTransformer.methods)
    finally:
        for name in placeholders:
            del sys.modules[name]
    rng = np.random.RandomState(7)
    out = {}
    for d in (5, 8, 100, 128):
        table = rng.normal(size=(40, d)).astype(np.float32)
        table[3] = 0.0  # zero rows exercise the cosine epsilon clamp
        table[4] = 1e-5
        src = rng.randint(0, 40, size=64).astype(np.int64)
        dst = rng.randint(0, 40, size=64).astype(np.int64)
        src[:4], dst[:4] = [3, 3, 4, 7], [3, 9, 4, 7]
        out[f"table_{d}"], out[f"src_{d}"], out[f"dst_{d}"] = table, src, dst
        for name, fn in methods.items():
            out[f"{name}_{d}"] = fn(table[src], table[dst])
    out["method_names"] = np.array(list(methods))
    return out


class StandInGraph:
    """The getters the reference's transformers call on an ensmallen.Graph
    (embedding_transformers/graph_transformer.py:150-243, node_transformer.py:186-190,
    edge_prediction_transformer.py:118-144, edge_label_prediction_transformer.py:125-229,
    node_label_prediction_transformer.py:84-150), over one entry of
    helpers.prediction_graph_specs() (default: transformer_graph_spec()): CSR edge order, type
    vocabularies in order of first appearance, None = unknown type."""

    UNKNOWN = 0xFFFFFFFF

    def __init__(self, spec=None):
        import numpy as np

        if spec is None:
            names, edges, edge_types, node_types = transformer_graph_spec()
            spec = dict(names=names, edges=edges, directed=False, edge_types=edge_types,
                        node_types=node_types)
        self.names = spec["names"]
        self.directed = spec["directed"]
        edge_types = spec["edge_types"]
        self.typed_edges = edge_types is not None
        labels = edge_types if self.typed_edges else [None] * len(spec["edges"])
        self.edge_vocab = list(dict.fromkeys(t for t in labels if t is not None))
        triples = set()
        for (a, b), t in zip(spec["edges"], labels):
            tid = self.UNKNOWN if t is None else self.edge_vocab.index(t)
            triples.add((a, b, tid))
            if not self.directed:
                triples.add((b, a, tid))
        keys = sorted(triples)
        self.src = np.array([k[0] for k in keys], dtype=np.uint32)
        self.dst = np.array([k[1] for k in keys], dtype=np.uint32)
        self.etype = np.array([k[2] for k in keys], dtype=np.uint32)
        self.multi = len({(k[0], k[1]) for k in keys}) != len(keys)
        node_types = spec["node_types"]
        self.typed_nodes = node_types is not None
        per_node = node_types if self.typed_nodes else [None] * len(self.names)
        per_node = [None if ts is None else (list(ts) if isinstance(ts, (list, tuple)) else [ts])
                    for ts in per_node]
        self.node_vocab = list(dict.fromkeys(t for ts in per_node if ts for t in ts))
        self.ntype = [None if not ts else
                      np.array(sorted(self.node_vocab.index(t) for t in ts), dtype=np.uint32)
                      for ts in per_node]
        self.upper = self.src <= self.dst

    def get_name(self): return "StandIn"
    def is_directed(self): return self.directed
    def is_multigraph(self): return self.multi
    def get_number_of_nodes(self): return len(self.names)
    def get_number_of_directed_edges(self): return len(self.src)
    def get_number_of_edges(self): return len(self.src)
    def get_node_names(self): return self.names
    def get_directed_source_node_ids(self): return self.src
    def get_directed_destination_node_ids(self): return self.dst
    def get_source_node_ids(self, directed=True):
        return self.src if directed or self.directed else self.src[self.upper]
    def get_destination_node_ids(self, directed=True):
        return self.dst if directed or self.directed else self.dst[self.upper]
    def get_directed_edge_node_names(self):
        return [(self.names[s], self.names[d]) for s, d in zip(self.src, self.dst)]
    def has_compatible_node_vocabularies(self, other):
        return self.names == other.names and self.typed_nodes == other.typed_nodes
    # node types
    def has_node_types(self): return self.typed_nodes
    def get_node_type_ids(self): return self.ntype
    def get_node_type_ids_from_node_id(self, i): return self.ntype[int(i)]
    def get_node_type_names_from_node_name(self, name):
        ids = self.ntype[self.names.index(name)]
        return None if ids is None else [self.node_vocab[i] for i in ids]
    def has_unknown_node_types(self): return self.typed_nodes and any(t is None for t in self.ntype)
    def has_known_node_types(self): return any(t is not None for t in self.ntype)
    def _node_label_counts(self):
        counts = [0] * len(self.node_vocab)
        for ids in self.ntype:
            for t in ([] if ids is None else ids):
                counts[int(t)] += 1
        return counts
    def has_homogeneous_node_types(self): return sum(c > 0 for c in self._node_label_counts()) == 1
    def has_singleton_node_types(self): return any(c == 1 for c in self._node_label_counts())
    def has_multilabel_node_types(self): return any(t is not None and len(t) > 1 for t in self.ntype)
    def get_one_hot_encoded_node_types(self):
        import numpy as np
        out = np.zeros((len(self.names), len(self.node_vocab)), dtype=bool)
        for i, ids in enumerate(self.ntype):
            if ids is not None:
                out[i, ids] = True
        return out
    def get_nodes_with_known_node_types_mask(self):
        import numpy as np
        return np.array([t is not None for t in self.ntype])
    # edge types
    def has_edge_types(self): return self.typed_edges
    def has_unknown_edge_types(self): return self.typed_edges and bool((self.etype == self.UNKNOWN).any())
    def has_known_edge_types(self): return self.typed_edges and bool((self.etype != self.UNKNOWN).any())
    def must_not_contain_unknown_edge_types(self):
        if self.has_unknown_edge_types():
            raise ValueError("unknown edge types")
    def must_not_be_multigraph(self):
        if self.multi:
            raise ValueError("multigraph")
    def get_imputed_directed_edge_type_ids(self, imputation_edge_type_id=0):
        import numpy as np
        return np.where(self.etype == self.UNKNOWN, np.uint32(imputation_edge_type_id), self.etype)
    def get_imputed_upper_triangular_edge_type_ids(self, imputation_edge_type_id=0):
        return self.get_imputed_directed_edge_type_ids(imputation_edge_type_id)[self.upper]
    def get_directed_edge_type_names(self):
        return [None if t == self.UNKNOWN else self.edge_vocab[t] for t in self.etype]
    def get_upper_triangular_edge_type_names(self):
        return [None if t == self.UNKNOWN else self.edge_vocab[t] for t in self.etype[self.upper]]
    def get_directed_edges_with_known_edge_types_mask(self): return self.etype != self.UNKNOWN
    def get_upper_triangular_known_edge_types_mask(self): return (self.etype != self.UNKNOWN)[self.upper]
    def get_directed_known_edge_type_ids(self): return self.etype[self.etype != self.UNKNOWN]
    def get_upper_triangular_known_edge_type_ids(self):
        upper = self.etype[self.upper]
        return upper[upper != self.UNKNOWN]
    def get_number_of_known_edge_types(self): return int((self.etype != self.UNKNOWN).sum())
    def get_edge_type_names_counts_hashmap(self):
        known = self.etype[self.etype != self.UNKNOWN]
        return {name: int((known == i).sum()) for i, name in enumerate(self.edge_vocab)}
    def has_homogeneous_edge_types(self):
        return sum(c > 0 for c in self.get_edge_type_names_counts_hashmap().values()) == 1
    def has_singleton_edge_types(self):
        once = 1 if self.directed else 2
        loops = self.src == self.dst
        for i in range(len(self.edge_vocab)):
            mine = self.etype == i
            edges = int(mine.sum()) if self.directed else (int(mine.sum()) + int((mine & loops).sum())) // once
            if edges == 1:
                return True
        return False


def transformer_fixture():
    """Runs helpers.transformer_cases against the reference's three transformer classes.  Their
    modules import `ensmallen` (for isinstance checks against Graph) and `userinput`; both are
    satisfied with placeholder modules whose Graph is StandInGraph for the duration of the call."""
    import types

    added = {}

    def put(name, module):
        if name not in sys.modules:
            added[name] = sys.modules[name] = module
        return sys.modules[name]

    ens = put("ensmallen", types.ModuleType("ensmallen"))
    ens.Graph = StandInGraph
    ens.express_measures = put("ensmallen.express_measures", types.ModuleType("ensmallen.express_measures"))
    put("userinput", types.ModuleType("userinput"))
    put("userinput.utils", types.ModuleType("userinput.utils")).must_be_in_set = lambda *a, **k: a[0]
    put("embiggen", types.ModuleType("embiggen"))
    put("embiggen.embedding_transformers", types.ModuleType("embiggen.embedding_transformers"))
    base = os.path.join(REF, "embiggen/embedding_transformers")
    try:
        loaded = {}
        for stem in ("node_transformer", "edge_transformer", "graph_transformer",
                     "edge_prediction_transformer", "edge_label_prediction_transformer",
                     "node_label_prediction_transformer"):
            full = f"embiggen.embedding_transformers.{stem}"
            spec = importlib.util.spec_from_file_location(full, os.path.join(base, stem + ".py"))
            loaded[stem] = put(full, importlib.util.module_from_spec(spec))
            spec.loader.exec_module(loaded[stem])
        T = types.SimpleNamespace(
            NodeTransformer=loaded["node_transformer"].NodeTransformer,
            EdgeTransformer=loaded["edge_transformer"].EdgeTransformer,
            GraphTransformer=loaded["graph_transformer"].GraphTransformer,
            EdgePredictionTransformer=loaded["edge_prediction_transformer"].EdgePredictionTransformer,
            EdgeLabelPredictionTransformer=loaded[
                "edge_label_prediction_transformer"].EdgeLabelPredictionTransformer,
            NodeLabelPredictionTransformer=loaded[
                "node_label_prediction_transformer"].NodeLabelPredictionTransformer)
        return transformer_cases(T, StandInGraph()), prediction_cases(T, StandInGraph)
    finally:
        for name in added:
            del sys.modules[name]


if __name__ == "__main__":
    import numpy as np

    for stem, cases in zip(("transformer_cases", "prediction_cases"), transformer_fixture()):
        np.savez_compressed(os.path.join(HERE, stem + ".npz"),
                            **{k: v for k, v in cases.items() if not isinstance(v, str)})
        with open(os.path.join(HERE, stem + ".json"), "w") as f:
            json.dump({k: v for k, v in cases.items() if isinstance(v, str)}, f, indent=1,
                      sort_keys=True)

    np.savez_compressed(os.path.join(HERE, "edge_embedding_cases.npz"), **edge_embedding_cases())
    with open(os.path.join(HERE, "embedding_result_cases.json"), "w") as f:
        json.dump(embedding_result_cases(), f, indent=1, sort_keys=True)
    with open(os.path.join(HERE, "api_defaults.json"), "w") as f:
        json.dump(api_defaults(), f, indent=1, sort_keys=True)
    print("written")

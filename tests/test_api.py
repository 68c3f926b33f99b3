"""Host-side drop-in contract of the embedders (no GPU needed).

Every expectation is anchored on the reference: constructor defaults / order, smoke-test
parameters, removed parameters, capability flags and schema types come from
tests/golden/api_defaults.json (extracted from the reference sources with ast by
tests/golden/make_reference_fixtures.py); the behaviours mirror the reference's own tests
tests/test_node_embedding_pipelines.py:83-105 (model recreation), tests/test_normalize_kwargs.py:10-31,
tests/test_abstract_model.py:124-138 (registry), tests/test_embed_graph_pipeline.py:59-94 and
tests/test_stub_model.py:7-9."""
import inspect
import json
import os

import numpy as np
import pytest

import embiggen_amd as E
from embiggen_amd import _lib
from embiggen_amd.utils.normalize_kwargs import SCHEMA
from helpers import GOLDEN

API = json.load(open(os.path.join(GOLDEN, "api_defaults.json")))
CLASSES = {
    "Node2VecSkipGramEnsmallen": E.Node2VecSkipGramEnsmallen,
    "Node2VecCBOWEnsmallen": E.Node2VecCBOWEnsmallen,
    "DeepWalkSkipGramEnsmallen": E.DeepWalkSkipGramEnsmallen,
    "DeepWalkCBOWEnsmallen": E.DeepWalkCBOWEnsmallen,
    "WalkletsSkipGramEnsmallen": E.WalkletsSkipGramEnsmallen,
    "WalkletsCBOWEnsmallen": E.WalkletsCBOWEnsmallen,
    "Node2VecGloVeEnsmallen": E.Node2VecGloVeEnsmallen,
    "DeepWalkGloVeEnsmallen": E.DeepWalkGloVeEnsmallen,
    "WalkletsGloVeEnsmallen": E.WalkletsGloVeEnsmallen,
}


@pytest.mark.parametrize("name", sorted(CLASSES))
def test_constructor_signature_matches_reference(name):
    sig = inspect.signature(CLASSES[name].__init__)
    params = [p for p in sig.parameters.values() if p.name != "self"]
    assert [p.name for p in params] == API[name]["init_order"]
    assert {p.name: p.default for p in params} == API[name]["init"]
    assert CLASSES[name].model_name() == API[name]["model_name"]


@pytest.mark.parametrize("name", sorted(CLASSES))
def test_capability_flags_match_reference(name):
    cls = CLASSES[name]
    flags = dict(API["Node2VecEnsmallen"]["flags"], **API["EnsmallenEmbedder"]["flags"])
    for method, value in flags.items():
        assert getattr(cls, method)() == value, method
    assert cls.smoke_test_parameters() == API["Node2VecEnsmallen"]["smoke_test_parameters"]
    model = cls()
    assert model.is_using_edge_weights() and not model.is_using_node_types()
    assert not model.is_using_edge_types()


@pytest.mark.parametrize("name", sorted(CLASSES))
def test_model_recreation_round_trip(name):
    model = CLASSES[name]()
    parameters = model.parameters()
    for removed in API[name]["removed"]:
        assert removed not in parameters
    second = CLASSES[name](**parameters)
    assert second.parameters() == parameters
    expected = {k: v for k, v in API[name]["init"].items()
                if k not in API[name]["removed"] + ["ring_bell", "enable_cache"]}
    assert parameters == expected
    smoke = model.into_smoke_test()
    for key, value in CLASSES[name].smoke_test_parameters().items():
        assert smoke.parameters()[key] == value
    assert model.consistent_hash() == second.consistent_hash() != smoke.consistent_hash()


@pytest.mark.parametrize("name", sorted(CLASSES))
def test_normalize_kwargs_round_trip(name):
    model = CLASSES[name]()
    CLASSES[name](**E.normalize_kwargs(model, model.parameters()))
    CLASSES[name](**E.normalize_kwargs(model, model.smoke_test_parameters()))


def test_schema_types_match_reference():
    for key, types in API["schema_types"].items():
        want = tuple([types] if isinstance(types, str) else types)
        assert SCHEMA[key] == want, key


def test_kwarg_coercion_and_errors():
    m = E.Node2VecSkipGramEnsmallen(embedding_size=np.int64(16), epochs=3.0,
                                    learning_rate="0.05", use_scale_free_distribution=np.bool_(0))
    p = m.parameters()
    assert p["embedding_size"] == 16 and type(p["embedding_size"]) is int
    assert p["epochs"] == 3 and type(p["epochs"]) is int
    assert p["learning_rate"] == "0.05" or p["learning_rate"] == 0.05
    assert p["use_scale_free_distribution"] is False
    with pytest.raises(NotImplementedError):  # unknown kwarg (normalize_kwargs.py:127-134)
        E.utils.normalize_kwargs(m, {"not_a_parameter": 1})
    with pytest.raises(TypeError):  # uncoercible (normalize_kwargs.py:117-125)
        E.Node2VecSkipGramEnsmallen(epochs=[3])
    with pytest.raises(NotImplementedError):  # int("many"): ValueError swallowed at :57-60, :66-71
        E.Node2VecSkipGramEnsmallen(epochs="many")
    with pytest.raises(TypeError):  # int(None): the reference rejects max_neighbours=None too
        E.Node2VecSkipGramEnsmallen(max_neighbours=None)
    with pytest.raises(ValueError):  # abstract_embedding_model.py:37-41
        E.Node2VecSkipGramEnsmallen(embedding_size=0)
    with pytest.raises(ValueError):  # stochastic model without seed (abstract_model.py:41-48)
        E.embedders.EnsmallenEmbedder.__init__(E.Node2VecSkipGramEnsmallen.__new__(
            E.Node2VecSkipGramEnsmallen), random_state=None, embedding_size=4)
    with pytest.raises(ValueError):  # f16 / f32 / f64 are the data types there are
        E.Node2VecSkipGramEnsmallen(dtype="bf16")
    assert E.Node2VecSkipGramEnsmallen(dtype="f16").parameters()["dtype"] == "f16"


def test_walklets_split_the_embedding_size_per_scale():
    """walklets.py:113 (embedding_size // window_size per scale) and :138-142 (parameters() reports
    the total again)."""
    m = E.WalkletsSkipGramEnsmallen(embedding_size=96, window_size=3)
    assert m.parameters()["embedding_size"] == 96 and m.parameters()["window_size"] == 3
    scales = m._model._scales
    assert [s.embedding_size for s in scales] == [32, 32, 32]
    assert [(s.window_size, s.min_distance) for s in scales] == [(1, 1), (2, 2), (3, 3)]
    abstract = json.load(open(os.path.join(GOLDEN, "api_defaults.json")))["WalkletsEnsmallen"]
    sig = inspect.signature(E.embedders.WalkletsEnsmallen.__init__)
    assert {p.name: p.default for p in sig.parameters.values() if p.name != "self"} == abstract["init"]


def test_set_random_state_reaches_the_engine_model():
    m = E.Node2VecCBOWEnsmallen()
    m.set_random_state(7)
    assert m.parameters()["random_state"] == 7 and m._model.random_state == 7


def test_registry():
    df = E.get_available_models_for_node_embedding()
    assert set(df.model_name) == {c.model_name() for c in CLASSES.values()}
    assert (df.library_name == "Ensmallen").all() and df.available.all()
    cls = E.AbstractEmbeddingModel.get_model_from_library("Node2Vec SkipGram")
    assert cls is E.Node2VecSkipGramEnsmallen
    cls = E.AbstractEmbeddingModel.get_model_from_library(
        "Node2Vec CBOW", task_name="Node Embedding", library_name="Ensmallen")
    assert cls is E.Node2VecCBOWEnsmallen
    assert len(E.AbstractModel.find_available_models("DeepWalk CBOW", "Node Embedding")) == 1
    for args in (("Unknown model", "Node Embedding"), ("Node2Vec CBOW", "Unknown task"),
                 ("", "Node Embedding"), ("Node2Vec CBOW", "")):
        with pytest.raises(ValueError):
            E.AbstractModel.find_available_models(*args)
    with pytest.raises(ValueError):
        E.AbstractEmbeddingModel.get_model_from_library("Node2Vec CBOW", library_name="Nope")


class _NotStochastic(E.AbstractEmbeddingModel):
    def __init__(self):
        super().__init__(embedding_size=100)

    @classmethod
    def smoke_test_parameters(cls):
        return dict(invalid_parameter=5)

    @classmethod
    def is_stocastic(cls):
        return False

    @classmethod
    def model_name(cls):
        return "TMP"

    @classmethod
    def library_name(cls):
        return "TMP"

    @classmethod
    def requires_nodes_sorted_by_decreasing_node_degree(cls):
        return False


def test_embed_graph_error_conventions(karate):
    """tests/test_embed_graph_pipeline.py:59-94 of the reference."""
    with pytest.raises(ValueError):
        E.embed_graph(karate, embedding_model=int)
    with pytest.raises(ValueError):  # instance + kwargs
        E.embed_graph(karate, embedding_model=E.Node2VecSkipGramEnsmallen(), embedding_size=10)
    with pytest.raises(ValueError):  # bad smoke-test parameters
        E.embed_graph(karate, embedding_model=_NotStochastic(), smoke_test=True)
    with pytest.raises(ValueError):  # _fit_transform not implemented -> wrapped
        E.embed_graph(karate, embedding_model=_NotStochastic())
    with pytest.raises(ValueError):  # unknown model name
        E.embed_graph(karate, embedding_model="No such model")
    with pytest.raises(ValueError):  # graph names need the network
        E.embed_graph("Cora", embedding_model="Node2Vec SkipGram", repository="linqs")


def test_graph_validation_errors(karate):
    model = E.Node2VecSkipGramEnsmallen(verbose=False)
    neg = E.CSRGraph.from_edge_list([0, 1], [1, 2], [1.0, -2.0], number_of_nodes=3)
    with pytest.raises(ValueError, match="negative edge weights"):
        model.fit_transform(neg)
    no_edges = E.CSRGraph.from_edge_list([], [], number_of_nodes=3)
    with pytest.raises(ValueError, match="does not have edges"):
        model.fit_transform(no_edges)
    empty = E.CSRGraph.from_edge_list([], [], number_of_nodes=0)
    with pytest.raises(ValueError, match="is empty"):
        model.fit_transform(empty)


def test_no_silent_cpu_fallback(karate):
    """Without a GPU the hot path must fail loudly, never compute elsewhere."""
    if _lib.device_count() > 0:
        pytest.skip("a GPU is visible")
    model = E.Node2VecSkipGramEnsmallen(verbose=False).into_smoke_test()
    with pytest.raises(RuntimeError, match="no CPU execution path"):
        model.fit_transform(karate)
    with pytest.raises(ValueError):  # embed_graph wraps it (graph_embedding_pipeline.py:94-107)
        E.embed_graph(karate, "Node2Vec SkipGram", smoke_test=True)


def test_missing_library_is_a_module_not_found(monkeypatch, tmp_path):
    """Counterpart of the reference's stub test (tests/test_stub_model.py:7-9): a missing backend
    raises ModuleNotFoundError with build instructions."""
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "libgn2v.so"))
    with pytest.raises(ModuleNotFoundError, match="no CPU fallback"):
        _lib.lib()


def test_node2vec_sequence_signature_and_shape_contract(karate):
    """Constructor defaults of the reference sequence (node2vec_sequence.py:14-27) and the batch
    geometry (:115-128); the batch content itself needs the GPU (tests/test_gpu_api.py)."""
    sig = inspect.signature(E.Node2VecSequence.__init__)
    ours = {p.name: p.default for p in sig.parameters.values()
            if p.name not in ("self", "graph", "device", "return_device_tensors")}
    assert ours == API["sequence_defaults"]
    seq = E.Node2VecSequence(karate, walk_length=20, batch_size=8, iterations=3, window_size=2)
    assert seq.sample_number == 34 and len(seq) == 5 and seq.steps_per_epoch == 5
    assert seq.number_of_skipgrams == 8 * 3 * 16
    with pytest.raises(ValueError):
        E.Node2VecSequence(karate, walk_length=8, window_size=4)


def test_cache_key_depends_on_graph_content(tmp_path, monkeypatch):
    """Two graphs with the same (default) name and the same parameters must not share a cache
    entry: the reference's `@Cache` hashes the graph argument
    (utils/abstract_models/abstract_embedding_model.py:91-95)."""
    import embiggen_amd as E

    g1 = E.CSRGraph.from_edge_list([0, 1, 2], [1, 2, 3], number_of_nodes=4)
    g2 = E.CSRGraph.from_edge_list([0, 1, 2, 3], [1, 2, 3, 4], number_of_nodes=5)
    g1_again = E.CSRGraph.from_edge_list([0, 1, 2], [1, 2, 3], number_of_nodes=4)
    assert g1.get_name() == g2.get_name()
    model = E.Node2VecSkipGramEnsmallen(embedding_size=4, enable_cache=True)
    p1, p2 = model._cache_path(g1, True), model._cache_path(g2, True)
    assert p1 != p2 and p1 == model._cache_path(g1_again, True)
    assert p1 != model._cache_path(g1, False)
    g1.get_node_names()  # materialising the default names must not move the key
    assert g1.content_digest() == g1_again.content_digest()
    named = E.CSRGraph.from_edge_list([0, 1, 2], [1, 2, 3], number_of_nodes=4,
                                      node_names=["a", "b", "c", "d"])
    assert named.content_digest() != g1.content_digest()


def test_gn2v_negatives_environment_reaches_the_reference_semantic_schedule(monkeypatch):
    """``GN2V_NEGATIVES=global``: the drop-in classes (which have no such kwarg: their signature is
    the reference's, node2vec_skipgram.py:9-36) train through the walk-ordered schedule -- every
    negative the endpoint of a uniform random edge of the whole graph (:101-102) -- instead of the
    block path's draw among the context's cell-mates.  A model that chose itself keeps its
    choice; anything but 'global' / 'cell' is refused."""
    from embiggen_amd import _lib, models

    m = E.Node2VecSkipGramEnsmallen(embedding_size=8, verbose=False)._model
    monkeypatch.delenv("GN2V_NEGATIVES", raising=False)
    assert not m.train_params().flags & (_lib.TRAIN_WALK_ORDERED | _lib.TRAIN_BLOCK_PATH)
    monkeypatch.setenv("GN2V_NEGATIVES", "global")
    assert m.train_params().flags & _lib.TRAIN_WALK_ORDERED
    assert models.SkipGram(embedding_size=8, block_path=True).train_params().flags \
        & _lib.TRAIN_BLOCK_PATH
    monkeypatch.setenv("GN2V_NEGATIVES", "cell")
    assert not m.train_params().flags & _lib.TRAIN_WALK_ORDERED
    monkeypatch.setenv("GN2V_NEGATIVES", "everywhere")
    with pytest.raises(ValueError, match="GN2V_NEGATIVES"):
        m.train_params()


def test_rounds_per_epoch_rule(monkeypatch):
    """Resident cells: 192 placements over a fit, 16 to 64 an epoch (csrc/handle.h
    rounds_per_epoch; the Python trainer's mirror)."""
    from embiggen_amd.distributed import rounds_per_epoch

    monkeypatch.delenv("GN2V_ROUNDS_PER_EPOCH", raising=False)
    assert [rounds_per_epoch(e) for e in (1, 2, 3, 4, 6, 12, 13, 30, 100)] == \
        [64, 64, 64, 48, 32, 16, 16, 16, 16]
    monkeypatch.setenv("GN2V_ROUNDS_PER_EPOCH", "5")
    assert rounds_per_epoch(30) == 5

"""GPU: the public embedder API end to end (reference behaviours: node2vec.py:93-112 result order
and DataFrame form, tests/test_node_embedding_pipelines.py:17-42 smoke run of every model)."""
import numpy as np
import pandas as pd
import pytest

import embiggen_amd as E

pytestmark = pytest.mark.gpu


def test_embed_graph_smoke_for_every_registered_model(karate):
    df = E.get_available_models_for_node_embedding()
    for _, row in df.iterrows():
        kwargs = {} if "Walklets" in row.model_name else {"verbose": False}
        res = E.embed_graph(karate, row.model_name, library_name=row.library_name,
                            smoke_test=True, **kwargs)
        tables = res.get_all_node_embedding()
        assert len(tables) == 2 and res.embedding_method_name == row.model_name
        for t in tables:
            assert isinstance(t, pd.DataFrame) and t.shape == (34, 5)
            assert list(t.index) == karate.get_node_names()
            assert np.isfinite(t.to_numpy()).all()


def test_result_order_skipgram_vs_cbow(karate):
    """SkipGram -> [central, contextual]; CBOW reversed so the input-side table comes first
    (node2vec.py:99-102).  With identical seeds both engines start from the same two tables."""
    kw = dict(embedding_size=8, epochs=0, verbose=False)
    sg = E.Node2VecSkipGramEnsmallen(**kw).fit_transform(karate, return_dataframe=False)
    cb = E.Node2VecCBOWEnsmallen(**kw).fit_transform(karate, return_dataframe=False)
    s0, s1 = sg.get_all_node_embedding()
    c0, c1 = cb.get_all_node_embedding()
    assert s0.dtype == np.float32 and s0.shape == (34, 8) and s0.flags.c_contiguous
    assert np.array_equal(s0, c1) and np.array_equal(s1, c0)  # zero epochs: just the init


@pytest.mark.parametrize("dtype,np_type,tol", [("f16", np.float16, 1e-3), ("f64", np.float64, 0)])
def test_result_data_types(karate, tmp_path, dtype, np_type, tol):
    """`dtype` (node2vec_skipgram.py:32,103-104): the engine computes in f32, the returned tables
    (and the memmaps of the *_embedding_path arguments) have the requested type and are the f32
    result converted."""
    kw = dict(embedding_size=8, epochs=1, walk_length=8, iterations=1, verbose=False)
    path = str(tmp_path / f"central_{dtype}.npy")
    ref = E.Node2VecSkipGramEnsmallen(**kw)
    got = E.Node2VecSkipGramEnsmallen(dtype=dtype, central_nodes_embedding_path=path, **kw)
    ref._model.deterministic = got._model.deterministic = True
    a = ref.fit_transform(karate, return_dataframe=False).get_all_node_embedding()
    b = got.fit_transform(karate, return_dataframe=False).get_all_node_embedding()
    for x, y in zip(a, b):
        assert y.dtype == np_type and y.shape == x.shape
        assert np.abs(y.astype(np.float64) - x.astype(np.float64)).max() <= tol
    assert np.load(path).dtype == np_type and np.array_equal(np.load(path), b[0])
    frames = got.fit_transform(karate).get_all_node_embedding()
    assert all(str(f.dtypes.iloc[0]) == np.dtype(np_type).name for f in frames)


def test_fit_is_reusable_and_seeded(karate):
    m = E.Node2VecSkipGramEnsmallen(embedding_size=8, epochs=1, walk_length=8, iterations=1,
                                    verbose=False)
    m._model.deterministic = True
    a = m.fit_transform(karate, return_dataframe=False).get_all_node_embedding()
    b = m.fit_transform(karate, return_dataframe=False).get_all_node_embedding()
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
    m.set_random_state(43)
    c = m.fit_transform(karate, return_dataframe=False).get_all_node_embedding()
    assert not np.array_equal(a[0], c[0])


def test_embedding_paths_are_memmapped(karate, tmp_path):
    cp, xp = str(tmp_path / "central.npy"), str(tmp_path / "ctx" / "contextual.npy")
    m = E.Node2VecSkipGramEnsmallen(embedding_size=8, epochs=1, walk_length=8, iterations=1,
                                    central_nodes_embedding_path=cp,
                                    contextual_nodes_embedding_path=xp, verbose=False)
    res = m.fit_transform(karate, return_dataframe=False).get_all_node_embedding()
    assert np.array_equal(np.load(cp), res[0]) and np.array_equal(np.load(xp), res[1])


def test_cache_round_trip(karate, tmp_path, monkeypatch):
    monkeypatch.chdir(tmp_path)
    m = E.Node2VecSkipGramEnsmallen(embedding_size=8, epochs=1, walk_length=8, iterations=1,
                                    enable_cache=True, verbose=False)
    a = m.fit_transform(karate, return_dataframe=False).get_all_node_embedding()
    b = m.fit_transform(karate, return_dataframe=False).get_all_node_embedding()
    assert np.array_equal(a[0], b[0])
    assert any(f.endswith(".pkl.gz") for _, _, fs in __import__("os").walk(tmp_path) for f in fs)


def test_weighted_graph_and_disconnected_warning():
    g = E.CSRGraph.from_edge_list([0, 1, 2], [1, 2, 0], [1.0, 2.0, 3.0], number_of_nodes=5)
    with pytest.warns(UserWarning, match="disconnected"):
        res = E.Node2VecSkipGramEnsmallen(embedding_size=4, epochs=1, walk_length=6,
                                          verbose=False).fit_transform(g)
    assert res.get_node_embedding_from_index(0).shape == (5, 4)


def test_node2vec_sequence_batches_match_oracle(karate, karate_oracle):
    """(contexts, words) batches of the sequence == the oracle's windows over the oracle's walks
    with the reference's seeding rule random_state + idx + elapsed_epochs
    (node2vec_sequence.py:190-203)."""
    from oracle import oracle as O

    seq = E.Node2VecSequence(karate, walk_length=20, batch_size=8, iterations=3, window_size=2,
                             return_weight=0.5, explore_weight=2.0, random_state=5)
    owp = O.WalkParams(20, 3, 0.5, 2.0, 100, 0)
    for idx, epochs in ((0, 0), (3, 0), (4, 1)):  # idx 4 wraps around the 34 sources
        seq.elapsed_epochs = epochs
        (((contexts, words),),) = seq[idx]
        assert contexts.shape == (8 * 3 * 16, 4) and contexts.dtype == np.int32
        assert words.shape == (8 * 3 * 16,) and words.dtype == np.int32
        seed, first = 5 + idx + epochs, (idx * 8) % 34
        ref = np.concatenate([O.walks(karate_oracle, owp, seed, 0, 2 * it * 34 + first, 8)
                              for it in range(3)])
        rc, rw = O.window_batch(ref, 2)
        assert np.array_equal(contexts, rc) and np.array_equal(words, rw)
        assert set(ref[:, 0].tolist()) == {(first + b) % 34 for b in range(8)}
    a = seq()
    assert a[0][0][0].shape == (384, 4)


@pytest.mark.parametrize("cls", [E.Node2VecSkipGramEnsmallen, E.Node2VecCBOWEnsmallen,
                                 E.Node2VecGloVeEnsmallen])
def test_wide_embeddings(karate, cls):
    """embedding_size in (512, 1024]: rows of 16 float4 per lane."""
    kw = {} if cls is E.Node2VecGloVeEnsmallen else {"iterations": 2}  # GloVe: one iteration
    m = cls(embedding_size=600, epochs=2, walk_length=16, verbose=False, **kw)
    tabs = m.fit_transform(karate, return_dataframe=False).get_all_node_embedding()
    assert all(t.shape == (34, 600) and np.isfinite(t).all() for t in tabs)
    init = cls(embedding_size=600, epochs=0, verbose=False).fit_transform(
        karate, return_dataframe=False).get_all_node_embedding()
    assert all(np.abs(a - b).max() > 1e-4 for a, b in zip(tabs, init))

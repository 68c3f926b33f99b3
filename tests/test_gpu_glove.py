"""GPU parity of the GloVe path: co-occurrence slots bit-exact, entries equal, the HIP SGD kernel
equal to the oracle in deterministic mode and on collision-free entries in every update mode."""
import numpy as np
import pytest
import torch

import embiggen_amd as E
from embiggen_amd import _lib, cooccurrence, models, ops
from helpers import link_auc, ring_of_cliques
from oracle import oracle as O

pytestmark = pytest.mark.gpu


def _u32(t):
    return t.cpu().numpy().view(np.uint32)


def _i64_as_u64(t):
    return t.cpu().numpy().view(np.uint64)


@pytest.mark.parametrize("window,min_dist,L", [(1, 1, 8), (5, 1, 40), (4, 4, 33), (3, 2, 128)])
def test_cooc_slots_bit_exact(karate, karate_oracle, window, min_dist, L):
    walks = ops.walks(karate, ops.walk_params(L, 3, 0.5, 2.0), 3, 0, 0, 102)
    keys, weights = ops.cooc_slots(walks, window, min_dist)
    okeys, oweights = O.cooc_slots(_u32(walks), window, min_dist)
    assert np.array_equal(_i64_as_u64(keys), okeys) and np.array_equal(_i64_as_u64(weights), oweights)


def test_cooc_slots_with_trapped_walks():
    rng = np.random.RandomState(0)
    g = E.CSRGraph.from_edge_list(rng.randint(0, 200, 500), rng.randint(0, 200, 500),
                                  number_of_nodes=220, directed=True)
    n = g.get_number_of_unique_source_nodes()
    walks = ops.walks(g, ops.walk_params(30, 1, 1.0, 1.0), 4, 0, 0, n)
    assert (_u32(walks) == _lib.SENTINEL).any()
    keys, weights = ops.cooc_slots(walks, 4)
    okeys, oweights = O.cooc_slots(_u32(walks), 4)
    assert np.array_equal(_i64_as_u64(keys), okeys) and np.array_equal(_i64_as_u64(weights), oweights)
    assert ops.cooc_slots(walks[:0], 4)[0].numel() == 0


def test_device_entries_equal_the_oracle(karate, karate_oracle):
    walks = ops.walks(karate, ops.walk_params(64, 2, 0.25, 4.0), 9, 0, 0, 68)
    keys, counts = cooccurrence.reduce_slots(*ops.cooc_slots(walks, 5))
    okeys, ocounts = O.cooc_reduce(*O.cooc_slots(_u32(walks), 5))
    assert np.array_equal(_i64_as_u64(keys), okeys) and np.array_equal(_i64_as_u64(counts), ocounts)
    rows, cols, logx, fx = cooccurrence.entries(keys, counts, 42, 0.75)
    orows, ocols, ologx, ofx = O.glove_entries(okeys, ocounts, 42, 0.75)
    assert np.array_equal(_u32(rows), orows) and np.array_equal(_u32(cols), ocols)
    assert np.allclose(logx.cpu().numpy(), ologx, rtol=1e-6, atol=1e-7)
    assert np.allclose(fx.cpu().numpy(), ofx, rtol=1e-6)


def _entries(karate_oracle, seed=5, window=4):
    walks = O.walks(karate_oracle, O.WalkParams(32, 2, 0.5, 2.0, 100, 0), seed, 0, 0, 68)
    return O.glove_entries(*O.cooc_reduce(*O.cooc_slots(walks, window)), seed, 0.75)


def _state(n, d, ld, seed):
    c, x = O.init_table(n, d, ld, seed, 0, 0.3), O.init_table(n, d, ld, seed, 1, 0.3)
    rng = np.random.RandomState(seed)
    return [c, x, (rng.normal(size=n) * 0.1).astype(np.float32),
            (rng.normal(size=n) * 0.1).astype(np.float32)]


def _run_gpu(graph, entries, state, d, lr, flags, passes=1):
    dev = [torch.from_numpy(a.view(np.int32) if a.dtype == np.uint32 else a).cuda() for a in entries]
    tensors = [torch.from_numpy(a.copy()).cuda() for a in state]
    for _ in range(passes):
        ops.glove_step(graph, *dev, *tensors, d, lr, flags)
    torch.cuda.synchronize()
    return [t.cpu().numpy() for t in tensors]


@pytest.mark.parametrize("d,ld", [(8, 8), (6, 8), (100, 128), (128, 128), (200, 224), (512, 512), (700, 704)])
def test_deterministic_kernel_equals_oracle(karate, karate_oracle, d, ld):
    entries = _entries(karate_oracle)
    assert len(entries[0]) > 500
    state = _state(34, d, ld, 3)
    got = _run_gpu(karate, entries, state, d, 0.05, _lib.TRAIN_DETERMINISTIC, passes=2)
    want = [a.copy() for a in state]
    for _ in range(2):
        O.glove_step(*entries, *want, d, 0.05)
    for g, w in zip(got, want):
        assert np.abs(g - w).max() < 2e-5
    assert (got[0][:, d:] == 0).all() and (got[1][:, d:] == 0).all()


@pytest.mark.parametrize("flags", [_lib.TRAIN_ATOMIC, _lib.TRAIN_WRITE_THROUGH, _lib.TRAIN_WRITE_BACK])
@pytest.mark.parametrize("d,ld", [(8, 8), (128, 128), (100, 128)])
def test_update_modes_on_collision_free_entries(flags, d, ld):
    """Every row and column appears once: the racy modes must equal the sequential oracle."""
    n = 5000
    rng = np.random.RandomState(1)
    g = E.CSRGraph.from_edge_list(np.arange(n - 1), np.arange(1, n), number_of_nodes=n)
    rows = rng.permutation(n).astype(np.uint32)
    cols = rng.permutation(n).astype(np.uint32)
    logx = -rng.uniform(0.1, 8.0, size=n).astype(np.float32)
    fx = rng.uniform(0.01, 1.0, size=n).astype(np.float32)
    state = _state(n, d, ld, 8)
    got = _run_gpu(g, (rows, cols, logx, fx), state, d, 0.05, flags)
    want = [a.copy() for a in state]
    O.glove_step(rows, cols, logx, fx, *want, d, 0.05)
    for a, b in zip(got, want):
        assert np.abs(a - b).max() < 1e-5


@pytest.mark.parametrize("flags", [_lib.TRAIN_ATOMIC, _lib.TRAIN_WRITE_THROUGH, _lib.TRAIN_WRITE_BACK])
@pytest.mark.parametrize("d,ld", [(8, 8), (128, 128), (100, 128)])
def test_record_fast_path_equals_the_round_schedule(flags, d, ld):
    """Records of 16 slots with a common row (some padded), every row and every column used once:
    the parallel kernel's record path must equal o_glove_step_rounds exactly."""
    n_rec, R = 700, O.GLOVE_RECORD
    rng = np.random.RandomState(2)
    n = n_rec * R
    g = E.CSRGraph.from_edge_list(np.arange(n - 1), np.arange(1, n), number_of_nodes=n)
    rows = np.repeat(rng.permutation(n)[:n_rec].astype(np.uint32), R)
    cols = rng.permutation(n).astype(np.uint32)
    logx = -rng.uniform(0.1, 8.0, size=n).astype(np.float32)
    fx = rng.uniform(0.01, 1.0, size=n).astype(np.float32)
    fill = rng.randint(1, R + 1, size=n_rec)  # valid slots per record
    pad = np.arange(R)[None, :] >= fill[:, None]
    cols[pad.ravel()] = O.SENTINEL
    logx[pad.ravel()] = 0
    fx[pad.ravel()] = 0
    state = _state(n, d, ld, 8)
    got = _run_gpu(g, (rows, cols, logx, fx), state, d, 0.05, flags)
    want = [a.copy() for a in state]
    O.glove_step_rounds(rows, cols, logx, fx, *want, d, 0.05)
    for a, b in zip(got, want):
        assert np.abs(a - b).max() < 1e-5
    assert np.abs(want[0] - state[0]).max() > 1e-3


def test_non_finite_entries_are_skipped(karate):
    rows, cols = np.array([1, 2, 3], np.uint32), np.array([4, 5, 6], np.uint32)
    logx = np.array([-1.0, -np.inf, -2.0], np.float32)
    fx = np.array([0.5, 0.5, np.nan], np.float32)
    state = _state(34, 8, 8, 2)
    got = _run_gpu(karate, (rows, cols, logx, fx), state, 8, 0.05, _lib.TRAIN_DETERMINISTIC)
    want = [a.copy() for a in state]
    O.glove_step(rows, cols, logx, fx, *want, 8, 0.05)
    for a, b in zip(got, want):
        assert np.isfinite(a).all() and np.abs(a - b).max() < 1e-6
    assert not np.array_equal(got[0][1], state[0][1]) and np.array_equal(got[0][2], state[0][2])


def test_model_fit_equals_the_oracle_pipeline(karate, karate_oracle):
    m = models.GloVe(embedding_size=12, random_state=7, epochs=5, walk_length=40, iterations=2,
                     window_size=3, return_weight=0.5, explore_weight=2.0, learning_rate=0.05,
                     learning_rate_decay=0.9, alpha=0.6, deterministic=True, verbose=False)
    central, contextual = m.fit_transform(karate)
    walks = O.walks(karate_oracle, O.WalkParams(40, 2, 0.5, 2.0, 100, 0), 7, 0, 0, 68)
    entries = O.glove_entries(*O.cooc_reduce(*O.cooc_slots(walks, 3)), 7, 0.6)
    ld = m.padded_size
    state = [O.init_table(34, 12, ld, 7, 0, 12 ** -0.5), O.init_table(34, 12, ld, 7, 1, 12 ** -0.5),
             np.zeros(34, np.float32), np.zeros(34, np.float32)]
    lr = np.float32(0.05)
    for _ in range(5):
        O.glove_step(*entries, *state, 12, float(lr))
        lr = np.float32(lr * np.float32(0.9))
    assert central.shape == (34, 12)
    assert m.last_stats["entries"] == int((entries[1] != O.SENTINEL).sum())
    assert np.abs(central - state[0][:, :12]).max() < 2e-5
    assert np.abs(contextual - state[1][:, :12]).max() < 2e-5


def test_batched_cooccurrence_equals_one_shot(karate):
    m = models.GloVe(embedding_size=8, walk_length=32, iterations=3, window_size=4, verbose=False)
    whole = m.cooccurrence_device(karate)
    m.SLOTS_PER_BATCH = 32 * 8 * 7  # 7 walks at a time
    parts = m.cooccurrence_device(karate)
    assert torch.equal(whole[0], parts[0]) and torch.equal(whole[1], parts[1])


def test_embedders_learn_communities_and_keep_the_contract():
    src, dst, n = ring_of_cliques(8, 8)
    g = E.CSRGraph.from_edge_list(src, dst, number_of_nodes=n)
    model = E.DeepWalkGloVeEnsmallen(embedding_size=16, walk_length=64, window_size=4, verbose=False)
    res = model.fit_transform(g, return_dataframe=False)
    central, contextual = res.get_all_node_embedding()
    assert central.shape == (n, 16) and central.dtype == np.float32 and np.isfinite(central).all()
    assert link_auc(g, central, contextual) > 0.9
    assert model.model_name() == "DeepWalk GloVe" and res.embedding_method_name == "DeepWalk GloVe"
    k = E.karate_club()
    frames = E.Node2VecGloVeEnsmallen(embedding_size=8, epochs=3, walk_length=32,
                                      verbose=False).fit_transform(k).get_all_node_embedding()
    assert len(frames) == 2 and frames[0].shape == (34, 8) and list(frames[0].index) == k.get_node_names()
    w = E.WalkletsGloVeEnsmallen(embedding_size=12, epochs=2, walk_length=32, window_size=3)
    tables = w.fit_transform(k, return_dataframe=False).get_all_node_embedding()
    assert len(tables) == 6 and all(t.shape == (34, 4) for t in tables)
    smoke = E.Node2VecGloVeEnsmallen().into_smoke_test().fit_transform(k, return_dataframe=False)
    assert smoke.get_all_node_embedding()[0].shape == (34, 5)
    out = E.embed_graph(k, "Node2Vec GloVe", embedding_size=8, epochs=2, walk_length=16, verbose=False)
    assert out.get_all_node_embedding()[0].shape == (34, 8)


def test_full_size_cooccurrence_properties():
    """BA 1 M nodes, walks of 128, window 5 (1.3 x 10^8 slots): the matrix is symmetric, its total
    is the closed form for full-length walks, keys are unique and ascending, and two halves merged
    equal the whole."""
    g = E.barabasi_albert(1_000_000, 10, 42)
    L, w, nw = 128, 5, 100_000
    walks = ops.walks(g, ops.walk_params(L, 1, 0.25, 4.0), 42, 0, 0, nw)
    assert bool((walks != -1).all())  # no traps in a BA graph: every walk is full length
    keys, counts = cooccurrence.reduce_slots(*ops.cooc_slots(walks, w))
    assert bool((keys[1:] > keys[:-1]).all())
    per_walk = sum(2 * (L - dist) * (((1 << 20) + dist // 2) // dist) for dist in range(1, w + 1))
    assert int(counts.sum()) == nw * per_walk
    swapped = ((keys & 0xFFFFFFFF) << 32) | ((keys >> 32) & 0xFFFFFFFF)
    order = torch.argsort(swapped)
    assert torch.equal(swapped[order], keys) and torch.equal(counts[order], counts)
    half = nw // 2
    acc = cooccurrence.Accumulator()
    acc.add(cooccurrence.reduce_slots(*ops.cooc_slots(walks[:half], w)))
    acc.add(cooccurrence.reduce_slots(*ops.cooc_slots(walks[half:], w)))
    merged = acc.result()
    assert torch.equal(merged[0], keys) and torch.equal(merged[1], counts)
    rows, cols, logx, fx = cooccurrence.entries(keys, counts, 42, 0.75)
    used = cols != -1
    assert float(logx.max()) == 0.0 and bool((fx[used] > 0).all()) and bool((fx <= 1).all())
    assert int(used.sum()) == keys.numel() and rows.numel() % cooccurrence.RECORD == 0
    assert rows.numel() < 1.15 * keys.numel()  # ~110 entries per row here: little padding


def test_parallel_schedules_reach_the_sequential_loss():
    """Racy schedules vs the deterministic one on the same entries: same loss level (BA 5 k nodes:
    collisions are real but not dominant, as in the SGNS equivalence test)."""
    n = 5000
    g = E.barabasi_albert(n, 5, seed=3)
    kw = dict(embedding_size=32, random_state=11, epochs=8, walk_length=30, iterations=1,
              window_size=4, return_weight=1.0, explore_weight=1.0, learning_rate=0.05,
              learning_rate_decay=0.95, verbose=False)
    ref = models.GloVe(deterministic=True, **kw)
    keys, counts = ref.cooccurrence_device(g)
    rows, cols, logx, fx = cooccurrence.entries(keys, counts, 11, 0.75)
    host = [t.cpu().numpy() for t in (rows, cols, logx, fx)]
    host[0], host[1] = host[0].view(np.uint32), host[1].view(np.uint32)
    zero = np.zeros(n, np.float32)  # biases are internal; compare the bias-free loss level

    def loss_of(model):
        c, x, _ = model.fit_transform_device(g)
        assert torch.isfinite(c).all() and torch.isfinite(x).all()
        return O.glove_loss(*host, np.ascontiguousarray(c.cpu().numpy()),
                            np.ascontiguousarray(x.cpu().numpy()), zero, zero, 32)

    want = loss_of(ref)
    for mode in ("atomic", "write_through", "write_back"):
        got = loss_of(models.GloVe(update_mode=mode, **kw))
        assert abs(got - want) < 0.05 * want, (mode, got, want)


def test_glove_step_argument_errors(karate):
    import ctypes as C

    L = _lib.lib()
    dg = karate.device_graph(0)
    t = torch.zeros((34, 8), dtype=torch.float32, device="cuda")
    b = torch.zeros(34, dtype=torch.float32, device="cuda")
    ids = torch.zeros(4, dtype=torch.int32, device="cuda")
    v = torch.zeros(4, dtype=torch.float32, device="cuda")
    io = _lib.GloveIO(ids.data_ptr(), ids.data_ptr(), v.data_ptr(), v.data_ptr(), t.data_ptr(),
                      t.data_ptr(), b.data_ptr(), b.data_ptr())
    assert L.gn2v_glove_step(dg.handle, C.byref(io), 4, 8, 6, 0.05, 0, None) != 0
    assert b"ld" in L.gn2v_last_error()
    assert L.gn2v_glove_step(dg.handle, C.byref(io), 4, 0, 8, 0.05, 0, None) != 0
    assert L.gn2v_glove_step(None, C.byref(io), 4, 8, 8, 0.05, 0, None) != 0
    assert L.gn2v_glove_step(dg.handle, C.byref(io), 0, 8, 8, 0.05, 0, None) == 0
    bad = _lib.GloveIO(None, ids.data_ptr(), v.data_ptr(), v.data_ptr(), t.data_ptr(),
                       t.data_ptr(), b.data_ptr(), b.data_ptr())
    assert L.gn2v_glove_step(dg.handle, C.byref(bad), 4, 8, 8, 0.05, 0, None) != 0
    assert b"NULL" in L.gn2v_last_error()

"""Shared test helpers (scenario tables, small graph builders, metrics)."""
import os
import sys
import warnings

import numpy as np
import pandas as pd

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def scenarios():
    """name -> kwargs builders, shared with tests/test_embedding_result.py (kept in sync by name)."""
    rng = np.random.RandomState(0)
    ok = rng.uniform(size=(10, 4))
    return {
        "node_single": dict(node_embeddings=ok),
        "node_list_two": dict(node_embeddings=[ok, ok + 1]),
        "edge_single": dict(edge_embeddings=ok),
        "node_type_single": dict(node_type_embeddings=ok),
        "edge_type_single": dict(edge_type_embeddings=ok),
        "node_dataframe": dict(node_embeddings=pd.DataFrame(ok)),
        "node_dataframe_nan": dict(node_embeddings=pd.DataFrame(np.full((3, 2), np.nan))),
        "nothing": dict(),
        "string_instead_of_array": dict(edge_type_embeddings="hu"),
        "empty_rows": dict(edge_type_embeddings=np.zeros((0, 4))),
        "all_nan": dict(edge_type_embeddings=np.full((10, 4), np.nan)),
        "one_nan": dict(node_embeddings=np.where(np.arange(40).reshape(10, 4) == 7, np.nan, ok)),
        "one_inf": dict(node_embeddings=np.where(np.arange(40).reshape(10, 4) == 7, np.inf, ok)),
        "all_zero": dict(node_embeddings=np.zeros((10, 4))),
        "list_with_bad_member": dict(node_embeddings=[ok, "hu"]),
    }


def probe(EmbeddingResult, kwargs):
    """Outcome record of constructing + poking an EmbeddingResult."""
    rec = {}
    with warnings.catch_warnings(record=True) as caught:
        warnings.simplefilter("always")
        try:
            r = EmbeddingResult(embedding_method_name="Test", **kwargs)
        except Exception as e:  # noqa: BLE001
            return {"construct": type(e).__name__}
        rec["construct"] = "ok"
        rec["warnings"] = len(caught)
    rec["name"] = r.embedding_method_name
    rec["number_of_embeddings"] = r.number_of_embeddings()
    rec["is_single"] = r.is_single_embedding()
    for getter in ("get_all_node_embedding", "get_all_edge_embedding",
                   "get_all_node_type_embeddings", "get_all_edge_type_embeddings"):
        try:
            rec[getter] = len(getattr(r, getter)())
        except Exception as e:  # noqa: BLE001
            rec[getter] = type(e).__name__
    for getter in ("get_node_embedding_from_index", "get_edge_embedding_from_index",
                   "get_node_type_embedding_from_index", "get_edge_type_embedding_from_index"):
        for index in (0, 1, 2):
            try:
                rec[f"{getter}[{index}]"] = list(getattr(r, getter)(index).shape)
            except TypeError:
                rec[f"{getter}[{index}]"] = "TypeError-or-ValueError"  # len(None) in the reference
            except Exception as e:  # noqa: BLE001
                rec[f"{getter}[{index}]"] = type(e).__name__
    dumped = r.dump()
    rec["dump_keys"] = sorted(dumped.keys())
    rec["load_roundtrip"] = EmbeddingResult.load(dumped).number_of_embeddings()
    rec["proxies_mean"] = bool(rec["is_single"] and hasattr(r, "mean"))
    rec["proxies_to_numpy"] = bool(hasattr(r, "to_numpy"))
    if rec["proxies_to_numpy"]:
        rec["to_numpy_shape"] = list(r.to_numpy().shape)
    return rec




# ---------------------------------------------------------------------- graphs & metrics
def ring_of_cliques(n_cliques: int, clique_size: int):
    """Edge list of `n_cliques` cliques joined in a ring (clear community structure)."""
    src, dst = [], []
    for c in range(n_cliques):
        base = c * clique_size
        for i in range(clique_size):
            for j in range(i + 1, clique_size):
                src.append(base + i)
                dst.append(base + j)
        nxt = ((c + 1) % n_cliques) * clique_size
        src.append(base)
        dst.append(nxt + 1)
    return np.array(src), np.array(dst), n_cliques * clique_size


def cosine_matrix(table: np.ndarray) -> np.ndarray:
    """Pairwise cosine similarity with the reference's epsilon clamp
    (embiggen/embedding_transformers/edge_transformer.py:242-267)."""
    norms = np.linalg.norm(table, axis=1, keepdims=True)
    prod = norms * norms.T
    prod[prod < 1e-6] = 1e-6
    return (table @ table.T) / prod


def adjacency(graph) -> np.ndarray:
    n = graph.get_number_of_nodes()
    a = np.zeros((n, n), dtype=bool)
    rp = graph.row_ptr.astype(np.int64)
    for u in range(n):
        a[u, graph.col_idx[rp[u]:rp[u + 1]]] = True
    return a


def link_auc(graph, central, contextual) -> float:
    """AUROC of the SkipGram score u_i . v_j (symmetrised) for edges vs non-edges."""
    from sklearn.metrics import roc_auc_score

    a = adjacency(graph)
    s = central @ contextual.T
    s = s + s.T
    iu = np.triu_indices(a.shape[0], 1)
    return float(roc_auc_score(a[iu], s[iu]))


def exact_second_order_probs(graph, prev: int, cur: int, return_weight: float,
                             explore_weight: float) -> np.ndarray:
    """Exact node2vec transition distribution out of `cur` given the previous node."""
    rp = graph.row_ptr.astype(np.int64)
    neigh = graph.col_idx[rp[cur]:rp[cur + 1]]
    prev_neigh = set(graph.col_idx[rp[prev]:rp[prev + 1]].tolist())
    w = np.array([
        return_weight if x == prev else (1.0 if int(x) in prev_neigh else explore_weight)
        for x in neigh
    ], dtype=np.float64)
    if graph.cumw is not None:
        ew = graph.get_directed_edge_weights()[rp[cur]:rp[cur + 1]].astype(np.float64)
        w = w * ew
    return neigh, w / w.sum()


def typed_karate():
    """Karate Club with two node types (one multi-label, a few unknown) and three edge types
    (symmetric: both directions of an edge carry the same type; some unknown)."""
    import embiggen_amd as E

    edges = np.loadtxt(os.path.join(ROOT, "embiggen_amd", "data", "karate.edges"), dtype=np.int64)
    node_types = [None if v % 11 == 5 else (["club", "officer"] if v in (0, 33) else
                                            ("mr_hi" if v < 17 else "club")) for v in range(34)]
    edge_types = [None if (a * 7 + b) % 13 == 0 else ("abc"[(a * 5 + b * 3) % 3])
                  for a, b in edges]
    return E.CSRGraph.from_edge_list(edges[:, 0], edges[:, 1], number_of_nodes=34,
                                     name="TypedKarate", node_types=node_types,
                                     edge_types=edge_types)


def exact_typed_probs(graph, prev, cur: int, return_weight: float, explore_weight: float,
                      change_node_type_weight: float = 1.0, change_edge_type_weight: float = 1.0):
    """Exact transition distribution out of `cur` (previous node `prev`, or None at the first
    step) on a simple graph with optional node / edge types: product of the second-order factor,
    change_node_type_weight when the neighbour's type differs from cur's, change_edge_type_weight
    when the edge's type differs from the type of the edge prev -> cur, and the edge weight."""
    rp = graph.row_ptr.astype(np.int64)
    lo, hi = rp[cur], rp[cur + 1]
    neigh = graph.col_idx[lo:hi]
    w = np.ones(len(neigh), dtype=np.float64)
    if prev is not None:
        prow = graph.col_idx[rp[prev]:rp[prev + 1]]
        prev_neigh = set(prow.tolist())
        w *= np.array([return_weight if x == prev else
                       (1.0 if int(x) in prev_neigh else explore_weight) for x in neigh])
        if graph.edge_type_ids is not None:
            ptype = graph.edge_type_ids[rp[prev] + int(np.flatnonzero(prow == cur)[0])]
            w *= np.where(graph.edge_type_ids[lo:hi] != ptype, change_edge_type_weight, 1.0)
    if graph.node_type_ids is not None:
        nt = graph.node_type_ids
        w *= np.where(nt[neigh.astype(np.int64)] != nt[cur], change_node_type_weight, 1.0)
    if graph.cumw is not None:
        w = w * graph.get_directed_edge_weights()[lo:hi].astype(np.float64)
    return neigh, w / w.sum()

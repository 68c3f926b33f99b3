"""CSRGraph: the ensmallen.Graph stand-in (getters used at
abstract_embedding_model.py:114-180, node2vec.py:105, graph_embedding_pipeline.py:91-92)."""
import networkx as nx
import numpy as np
import pytest

import embiggen_amd as E


def test_karate_fixture(karate):
    assert karate.get_number_of_nodes() == 34
    assert karate.get_number_of_directed_edges() == 156
    assert karate.has_nodes() and karate.has_edges() and not karate.is_directed()
    assert not karate.has_node_types() and not karate.has_edge_types()
    assert not karate.has_edge_weights() and not karate.has_negative_edge_weights()
    assert not karate.has_disconnected_nodes()
    assert karate.get_number_of_unique_source_nodes() == 34 and karate.sources is None
    ref = nx.karate_club_graph()
    assert sorted(karate.get_node_degrees()) == sorted(d for _, d in ref.degree())
    assert karate.get_node_names()[:3] == ["0", "1", "2"]


def test_csr_is_sorted_unique_and_symmetric():
    g = E.CSRGraph.from_edge_list([3, 0, 0, 1, 1], [0, 1, 1, 0, 2], number_of_nodes=5)
    rp, ci = g.row_ptr.astype(int), g.col_idx
    assert rp.tolist() == [0, 2, 4, 5, 6, 6]
    assert ci.tolist() == [1, 3, 0, 2, 1, 0]
    assert g.has_disconnected_nodes() and g.get_number_of_disconnected_nodes() == 1
    assert g.sources.tolist() == [0, 1, 2, 3]
    assert np.array_equal(g.get_cumulative_node_degrees(), g.row_ptr[1:])


def test_weighted_graph_prefix_sums():
    g = E.CSRGraph.from_edge_list([0, 0, 1], [1, 2, 2], [1.0, 3.0, 0.5], number_of_nodes=3)
    assert g.has_edge_weights() and not g.has_negative_edge_weights()
    assert g.col_idx.tolist() == [1, 2, 0, 2, 0, 1]
    assert np.allclose(g.get_directed_edge_weights(), [1, 3, 1, 0.5, 3, 0.5])
    assert np.allclose(g.cumw, [1, 4, 1, 1.5, 3, 3.5])
    dup = E.CSRGraph.from_edge_list([0, 0], [1, 1], [1.0, 2.0], number_of_nodes=2)
    assert np.allclose(dup.get_directed_edge_weights(), [3, 3])


def test_from_networkx_and_sorting():
    ref = nx.barabasi_albert_graph(200, 3, seed=1)
    g = E.CSRGraph.from_networkx(ref, name="ba")
    assert g.get_name() == "ba" and g.get_number_of_directed_edges() == 2 * ref.number_of_edges()
    assert not g.has_nodes_sorted_by_decreasing_outbound_node_degree()
    s = g.sort_by_decreasing_outbound_node_degree()
    assert s.has_nodes_sorted_by_decreasing_outbound_node_degree()
    assert sorted(s.get_node_degrees()) == sorted(g.get_node_degrees())
    # relabelling keeps the edge set
    old_of_new = [int(n) for n in s.get_node_names()]
    a = nx.Graph((old_of_new[u], old_of_new[int(v)]) for u in range(200)
                 for v in s.col_idx[int(s.row_ptr[u]):int(s.row_ptr[u + 1])])
    assert nx.utils.graphs_equal(a, nx.Graph(ref.edges())) or set(map(frozenset, a.edges())) == set(
        map(frozenset, ref.edges()))


def test_invalid_inputs():
    with pytest.raises(ValueError):
        E.CSRGraph.from_edge_list([0], [5], number_of_nodes=3)
    with pytest.raises(ValueError):
        E.CSRGraph.from_csr([0, 1], [0, 1])
    with pytest.raises(ValueError):
        E.CSRGraph.from_edge_list([0, 1], [1], number_of_nodes=3)


def test_from_ensmallen_adapter():
    """Duck-typed ensmallen.Graph exposing the CSR getters of pecanpy_embedders/node2vec.py:144-163."""

    class Fake:
        def get_number_of_nodes(self): return 3
        def get_cumulative_node_degrees(self): return np.array([1, 3, 4])
        def get_directed_destination_node_ids(self): return np.array([1, 0, 2, 1])
        def has_edge_weights(self): return False
        def get_node_names(self): return ["a", "b", "c"]
        def get_name(self): return "fake"
        def is_directed(self): return False

    g = E.CSRGraph.from_ensmallen(Fake())
    assert g.row_ptr.tolist() == [0, 1, 3, 4] and g.get_node_names() == ["a", "b", "c"]

"""CPU tests of the GIN oracle (no GPU): cross-check against the independent NumPy restatement,
against the committed golden vectors, and the integer tables of load_graph."""
import os

import numpy as np
import pytest

from flowgnn_amd import graphpack as gp, weights
from tests import numpy_ref

GOLDEN = os.path.join(os.path.dirname(__file__), "golden", "gin_molhiv64.npz")
REF_GIN = "/root/reference/GIN"


def test_oracle_matches_numpy_float64(oracle, gin_weights):
    b = gp.synth_molhiv_batch(48, seed=11)
    out, hd = oracle.gin_forward(b, [gin_weights], dump_h=True)
    ref, hs = numpy_ref.gin_forward(b, gin_weights, return_h=True)
    # float32 oracle vs float64 restatement: tolerance of SURVEY 8c
    assert np.allclose(out, ref, rtol=1e-4, atol=1e-4), np.abs(out - ref).max()
    assert np.allclose(hd, hs, rtol=1e-4, atol=1e-4), np.abs(hd - hs).max()


def test_oracle_openmp_identical(oracle, gin_weights):
    b = gp.synth_molhiv_batch(40, seed=3)
    a = oracle.gin_forward(b, [gin_weights], nthreads=1)
    c = oracle.gin_forward(b, [gin_weights], nthreads=4)
    assert np.array_equal(a, c)


def test_oracle_golden_vectors(oracle, gin_weights):
    z = np.load(GOLDEN)
    b = gp.GraphBatch(z["nums_of_nodes"], z["nums_of_edges"], z["node_feature"], z["edge_list"], z["edge_attr"])
    out, hd = oracle.gin_forward(b, [gin_weights], dump_h=True)
    assert np.array_equal(out, z["logits_synth_weights"])  # same code, same machine class: bit-exact
    n4 = int(b.nums_of_nodes[:4].sum())
    assert np.array_equal(hd[:, :n4], z["h_first4_graphs"])


@pytest.mark.skipif(not os.path.isdir(REF_GIN), reason="reference weights not on this machine")
def test_oracle_golden_real_weights(oracle):
    z = np.load(GOLDEN)
    b = gp.GraphBatch(z["nums_of_nodes"], z["nums_of_edges"], z["node_feature"], z["edge_list"], z["edge_attr"])
    w = weights.load_gin_weights(REF_GIN)
    out = oracle.gin_forward(b, [w])
    assert np.array_equal(out, z["logits_reference_weights"])
    # independent restatement with the shipped weights as well
    assert np.allclose(out, numpy_ref.gin_forward(b, w), rtol=1e-4, atol=1e-4)


@pytest.mark.skipif(not os.path.isdir(REF_GIN), reason="reference weights not on this machine")
def test_reference_weight_files_have_expected_sizes():
    for k, (f, shp) in weights.GIN_FILES.items():
        assert os.path.getsize(os.path.join(REF_GIN, f)) == 4 * int(np.prod(shp)), f


def test_load_graph_tables(oracle):
    """Integer bookkeeping of GIN/src/load_inputs.cc:87-172 on a hand-checkable graph."""
    el = np.array([[0, 1], [1, 0], [2, 1], [1, 2], [0, 1], [3, 1], [1, 5], [5, 1]], dtype=np.int32)
    ea = np.arange(24, dtype=np.int32).reshape(8, 3) % 2
    t = oracle.gin_load_graph(el, ea, 6)
    assert t["degree_table"].tolist() == [2, 3, 1, 1, 0, 1]            # out-degrees
    assert t["num_of_edges_per_pe"].tolist() == [1, 6, 1, 0]           # bank = v % 4
    # PE 1 holds destinations 1 and 5: sources ascending, ties in input order
    assert t["neighbor_tables"][1, :6].tolist() == [0, 0, 1, 0, 0, 0]  # v / 4
    assert t["degree_tables"][1].tolist() == [2, 1, 1, 1, 0, 1]


def test_reload_weights_selects_weight_set(oracle, gin_weights):
    b = gp.synth_molhiv_batch(6, seed=5)
    w2 = weights.synth_gin_weights(seed=8)
    rw = np.array([1, 0, 0, 1, 0, 0], dtype=np.int32)
    out = oracle.gin_forward(b, [gin_weights, w2], reload_weights=rw)
    a = oracle.gin_forward(b.slice(0, 3), [gin_weights])
    c = oracle.gin_forward(b.slice(3, 6), [w2])
    assert np.array_equal(out, np.concatenate([a, c]))


def test_virtual_node_augmentation_shapes():
    b = gp.synth_molhiv_batch(5, seed=2)
    v = gp.add_virtual_nodes(b)
    assert np.array_equal(v.nums_of_nodes, b.nums_of_nodes + 1)
    assert np.array_equal(v.nums_of_edges, b.nums_of_edges + 2 * b.nums_of_nodes)
    n0, e0 = int(b.nums_of_nodes[0]), int(b.nums_of_edges[0])
    assert v.edge_list[e0].tolist() == [0, n0] and v.edge_list[e0 + 1].tolist() == [n0, 0]
    assert (v.node_feature[n0] == 0).all()


def test_pack_roundtrip(tmp_path):
    b = gp.synth_molhiv_batch(7, seed=9)
    gp.write_pack(b, str(tmp_path))
    r = gp.read_pack(str(tmp_path))
    for f in ("nums_of_nodes", "nums_of_edges", "node_feature", "edge_list", "edge_attr"):
        assert np.array_equal(getattr(b, f), getattr(r, f)), f
    with open(tmp_path / "graph_info" / "g1_info.txt") as fh:
        assert fh.read() == f"{b.nums_of_nodes[0]}\n{b.nums_of_edges[0]}"

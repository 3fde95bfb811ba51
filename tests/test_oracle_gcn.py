"""CPU tests of the GCN oracle: independent NumPy restatement, golden vectors, the .bin layout."""
import os

import numpy as np
import pytest

from flowgnn_amd import graphpack as gp, weights
from tests import numpy_ref

GOLDEN = os.path.join(os.path.dirname(__file__), "golden", "gcn_molpcba48.npz")
REF = "/root/reference/GCN"


def directed_variant(b):
    """Drop some reverse edges so that nodes without out-edges exist (degree_inv_sqrt stays 0 there)."""
    keep = np.ones(b.total_edges, bool)
    keep[1::6] = False
    eo = b.edge_offsets()
    ne = np.add.reduceat(keep.astype(np.int64), eo[:-1]).astype(np.int32)
    return gp.GraphBatch(b.nums_of_nodes, ne, b.node_feature, b.edge_list[keep], b.edge_attr[keep])


def test_oracle_matches_numpy_float64(oracle):
    w = weights.synth_gcn_weights(seed=7)
    for b in (gp.synth_molpcba_batch(40, seed=11), directed_variant(gp.synth_molpcba_batch(24, seed=12))):
        out, xd = oracle.gcn_forward(b, [w], dump_h=True)
        ref, xs = numpy_ref.gcn_forward(b, w, return_x=True)
        assert np.allclose(out, ref, rtol=1e-4, atol=1e-4), np.abs(out - ref).max()
        assert np.allclose(xd, xs, rtol=1e-4, atol=2e-4), np.abs(xd - xs).max()


def test_oracle_golden_vectors(oracle):
    z = np.load(GOLDEN)
    b = gp.GraphBatch(z["nums_of_nodes"], z["nums_of_edges"], z["node_feature"], z["edge_list"], z["edge_attr"])
    out = oracle.gcn_forward(b, [weights.synth_gcn_weights(seed=7)])
    assert np.array_equal(out, z["logits_synth_weights"])


@pytest.mark.skipif(not os.path.isdir(REF), reason="reference weights not on this machine")
def test_reference_weights(oracle):
    assert os.path.getsize(os.path.join(REF, weights.GCN_FILE)) == 4 * 76906
    w = weights.load_gcn_weights(REF)
    assert (w["bn_var"] > 0).all()  # a wrong BN offset would land on means/biases
    z = np.load(GOLDEN)
    b = gp.GraphBatch(z["nums_of_nodes"], z["nums_of_edges"], z["node_feature"], z["edge_list"], z["edge_attr"])
    out = oracle.gcn_forward(b, [w])
    assert np.array_equal(out, z["logits_reference_weights"])
    assert np.allclose(out, numpy_ref.gcn_forward(b, w), rtol=1e-4, atol=1e-4)


def test_bin_roundtrip(tmp_path):
    w = weights.synth_gcn_weights(seed=3)
    weights.save_gcn_weights(w, str(tmp_path))
    r = weights.load_gcn_weights(str(tmp_path))
    for k in w:
        assert np.array_equal(w[k], r[k]), k

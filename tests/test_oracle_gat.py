"""CPU tests of the GAT oracle: independent NumPy restatement, golden vectors, the weight files, the
reference's feature-offset quirk."""
import os

import numpy as np
import pytest

from flowgnn_amd import graphpack as gp, weights
from tests import numpy_ref
from tests.test_oracle_gcn import directed_variant

GOLDEN = os.path.join(os.path.dirname(__file__), "golden", "gat_molhiv48.npz")
REF = "/root/reference/GAT"


def batches():
    return (gp.synth_molhiv_batch(40, seed=11), directed_variant(gp.synth_molhiv_batch(12, seed=12)),
            gp.synth_hep10k_batch(6, seed=4, with_eigen=False))


def test_oracle_matches_numpy_float64(oracle):
    w = weights.synth_gat_weights(seed=7)
    for b in batches():
        out, hd = oracle.gat_forward(b, [w], dump_h=True)
        ref, hs = numpy_ref.gat_forward(b, w, return_h=True)
        s = max(1.0, float(np.abs(hs).max()))
        assert np.isfinite(out).all()
        assert np.allclose(hd, hs, rtol=1e-4, atol=1e-4 * s), np.abs(hd - hs).max()
        assert np.allclose(out, ref, rtol=1e-4, atol=1e-4 * s), np.abs(out - ref).max()


def test_feature_offset_quirk(oracle):
    """GAT_compute.cc:72: without the per-graph offset every graph reads the first rows of the batch."""
    w = weights.synth_gat_weights(seed=7)
    b = gp.synth_molhiv_batch(6, seed=3)
    quirk = oracle.gat_forward(b, [w], feature_offset_quirk=True)
    fixed = oracle.gat_forward(b, [w], feature_offset_quirk=False)
    assert quirk[0] == fixed[0] and not np.array_equal(quirk[1:], fixed[1:])
    # equivalent formulation: overwrite each graph's features with the batch's leading rows
    no = b.node_offsets()
    nf = b.node_feature.copy()
    for g in range(b.num_graphs):
        nf[no[g]:no[g + 1]] = b.node_feature[: b.nums_of_nodes[g]]
    b2 = gp.GraphBatch(b.nums_of_nodes, b.nums_of_edges, nf, b.edge_list, b.edge_attr)
    assert np.array_equal(oracle.gat_forward(b2, [w]), quirk)


def test_oracle_golden_vectors(oracle):
    z = np.load(GOLDEN)
    b = gp.GraphBatch(z["nums_of_nodes"], z["nums_of_edges"], z["node_feature"], z["edge_list"], z["edge_attr"])
    assert np.array_equal(oracle.gat_forward(b, [weights.synth_gat_weights(seed=7)]), z["logits_synth_weights"])


@pytest.mark.skipif(not os.path.isdir(REF), reason="reference weights not on this machine")
def test_reference_weights(oracle):
    w = weights.load_gat_weights(REF)
    assert os.path.getsize(os.path.join(REF, "gat_ep1_linear_proj_weight_1_layer5.bin")) == 4 * 4 * 4 * 16 * 4 * 16
    assert os.path.getsize(os.path.join(REF, "gat_ep1_skip_proj_weight_0_layer5.bin")) == 4 * 4 * 16 * 9
    z = np.load(GOLDEN)
    b = gp.GraphBatch(z["nums_of_nodes"], z["nums_of_edges"], z["node_feature"], z["edge_list"], z["edge_attr"])
    out = oracle.gat_forward(b, [w])
    assert np.array_equal(out, z["logits_reference_weights"], equal_nan=True)
    ref = numpy_ref.gat_forward(b, w)
    ok = np.isfinite(ref)
    assert np.allclose(out[ok], ref[ok], rtol=1e-3, atol=1e-3)


def test_bin_roundtrip(tmp_path):
    w = weights.synth_gat_weights(seed=3)
    weights.save_gat_weights(w, str(tmp_path))
    r = weights.load_gat_weights(str(tmp_path))
    for k in w:
        assert np.array_equal(np.asarray(w[k]), np.asarray(r[k])), k

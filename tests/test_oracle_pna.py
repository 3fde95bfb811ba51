"""CPU tests of the PNA oracle: independent NumPy restatement, golden vectors, the .bin layout."""
import os

import numpy as np
import pytest

from flowgnn_amd import graphpack as gp, weights
from tests import numpy_ref
from tests.test_oracle_gcn import directed_variant

GOLDEN = os.path.join(os.path.dirname(__file__), "golden", "pna_hep24.npz")
REF = "/root/reference/PNA"


def batches():
    # kNN graphs (in-degree 16, out-degree varies), molecules, and a directed variant with in-degree-0 nodes
    return (gp.synth_hep10k_batch(12, seed=11, with_eigen=False), gp.synth_molhiv_batch(20, seed=3),
            directed_variant(gp.synth_molhiv_batch(12, seed=12)))


def test_oracle_matches_numpy_float64(oracle):
    w = weights.synth_pna_weights(seed=7)
    for b in batches():
        out, hd = oracle.pna_forward(b, [w], dump_h=True)
        ref, hs = numpy_ref.pna_forward(b, w, return_h=True)
        assert np.allclose(out, ref, rtol=1e-4, atol=1e-4), np.abs(out - ref).max()
        # std = sqrt(Q/n - mean^2) cancels in fp32: tolerance relative to the activation scale
        assert np.allclose(hd, hs, rtol=1e-4, atol=1e-4 * max(1.0, np.abs(hs).max())), np.abs(hd - hs).max()


def test_oracle_golden_vectors(oracle):
    z = np.load(GOLDEN)
    b = gp.GraphBatch(z["nums_of_nodes"], z["nums_of_edges"], z["node_feature"], z["edge_list"], z["edge_attr"])
    assert np.array_equal(oracle.pna_forward(b, [weights.synth_pna_weights(seed=7)]), z["logits_synth_weights"])


@pytest.mark.skipif(not os.path.isdir(REF), reason="reference weights not on this machine")
def test_reference_weights(oracle):
    assert os.path.getsize(os.path.join(REF, weights.PNA_FILE)) == 4 * 325441
    w = weights.load_pna_weights(REF)
    z = np.load(GOLDEN)
    b = gp.GraphBatch(z["nums_of_nodes"], z["nums_of_edges"], z["node_feature"], z["edge_list"], z["edge_attr"])
    out = oracle.pna_forward(b, [w])
    assert np.array_equal(out, z["logits_reference_weights"])
    assert np.allclose(out, numpy_ref.pna_forward(b, w), rtol=1e-4, atol=1e-4)


def test_bin_roundtrip(tmp_path):
    w = weights.synth_pna_weights(seed=3)
    weights.save_pna_weights(w, str(tmp_path))
    r = weights.load_pna_weights(str(tmp_path))
    for k in w:
        assert np.array_equal(np.asarray(w[k]), np.asarray(r[k])), k

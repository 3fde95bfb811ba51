"""CPU tests of the DGN oracle: independent NumPy restatement, golden vectors, the .bin layout, the eig text format."""
import os

import numpy as np
import pytest

from flowgnn_amd import graphpack as gp, weights
from tests import numpy_ref

GOLDEN = os.path.join(os.path.dirname(__file__), "golden", "dgn_hep24.npz")
REF = "/root/reference/DGN"


def with_eigen(b, seed):
    rng = np.random.default_rng(seed)
    b.node_eigen = rng.uniform(-0.5, 0.5, (b.total_nodes, 4)).astype(np.float32)
    return b


def batches():
    return (gp.synth_hep10k_batch(12, seed=11), with_eigen(gp.synth_molhiv_batch(20, seed=3), 1))


def from_npz(z):
    return gp.GraphBatch(z["nums_of_nodes"], z["nums_of_edges"], z["node_feature"], z["edge_list"], z["edge_attr"], z["node_eigen"])


def test_oracle_matches_numpy_float64(oracle):
    w = weights.synth_dgn_weights(seed=7)
    for b in batches():
        out, hd = oracle.dgn_forward(b, [w], dump_h=True)
        ref, hs = numpy_ref.dgn_forward(b, w, return_h=True)
        s = max(1.0, float(np.abs(hs).max()))
        assert np.allclose(hd, hs, rtol=1e-4, atol=1e-4 * s), np.abs(hd - hs).max()
        assert np.allclose(out, ref, rtol=1e-4, atol=1e-4 * s), np.abs(out - ref).max()


def test_oracle_golden_vectors(oracle):
    z = np.load(GOLDEN)
    assert np.array_equal(oracle.dgn_forward(from_npz(z), [weights.synth_dgn_weights(seed=7)]), z["logits_synth_weights"])


@pytest.mark.skipif(not os.path.isdir(REF), reason="reference weights not on this machine")
def test_reference_weights(oracle):
    assert os.path.getsize(os.path.join(REF, weights.DGN_FILE)) == 4 * 104051
    w = weights.load_dgn_weights(REF)
    z = np.load(GOLDEN)
    out = oracle.dgn_forward(from_npz(z), [w])
    assert np.array_equal(out, z["logits_reference_weights"])
    ref = numpy_ref.dgn_forward(from_npz(z), w)
    assert np.allclose(out, ref, rtol=1e-3, atol=1e-3), np.abs(out - ref).max()


def test_bin_and_eig_roundtrip(tmp_path):
    w = weights.synth_dgn_weights(seed=3)
    weights.save_dgn_weights(w, str(tmp_path))
    r = weights.load_dgn_weights(str(tmp_path))
    for k in w:
        assert np.array_equal(np.asarray(w[k]), np.asarray(r[k])), k
    b = gp.synth_hep10k_batch(3, seed=2)
    gp.write_pack(b, str(tmp_path / "graphs"), eig_dir=str(tmp_path / "eig"))
    rb = gp.read_pack(str(tmp_path / "graphs"), eig_dir=str(tmp_path / "eig"))
    assert np.array_equal(b.edge_list, rb.edge_list)
    assert np.allclose(b.node_eigen, rb.node_eigen, rtol=1e-4, atol=1e-7)  # "%.4e" text

"""GPU parity tests for DGN (BASELINE config 5): HIP path through the C ABI vs the CPU oracle.
Tolerance (tests/parity.py): |gpu - oracle| <= 1e-4 * (scale + |oracle|), scale = the oracle's own largest activation (or logit where
no activations were dumped), measured per comparison (the directional term divides by sum |w_e|, which can be small)."""
import os

import numpy as np
import pytest

from flowgnn_amd import Engine, FlowGNNError, compute_graphs, graphpack as gp, weights
from tests.parity import assert_close, oracle_scale
from tests.test_oracle_dgn import with_eigen, from_npz

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(__file__), "golden", "dgn_hep24.npz")


@pytest.fixture(scope="module")
def w():
    return weights.synth_dgn_weights(seed=7)


@pytest.fixture(scope="module")
def eng(w):
    e = Engine("DGN", device=0)
    e.set_weights(w)
    yield e
    e.close()


def test_forward_matches_oracle(eng, oracle, w):
    for b in (gp.synth_hep10k_batch(48, seed=31), with_eigen(gp.synth_molhiv_batch(100, seed=3), 1)):
        got = eng.forward(b)
        want, hd = oracle.dgn_forward(b, [w], dump_h=True, nthreads=8)
        scale = oracle_scale(hd)
        assert_close(eng.final_h(), hd[4], scale, what="h_4")
        assert_close(got, want, scale, what="logits")


def test_golden_vectors(eng):
    z = np.load(GOLDEN)
    assert_close(eng.forward(from_npz(z)), z["logits_synth_weights"], what="golden logits")  # relative to the logits themselves


def test_entry_point_bin_loader_and_edge_cases(tmp_path, oracle, w):
    b = gp.synth_hep10k_batch(7, seed=5)
    w2 = weights.synth_dgn_weights(seed=8)
    rw = np.array([1, 0, 0, 1, 0, 0, 0], np.int32)
    want, hd = oracle.dgn_forward(b, [w2], dump_h=True)
    assert_close(compute_graphs("DGN", b, [w, w2], rw), oracle.dgn_forward(b, [w, w2], reload_weights=rw), oracle_scale(hd), what="two weight sets")
    weights.save_dgn_weights(w2, str(tmp_path))
    e = Engine("DGN", device=0)
    e.load_weights_dir(str(tmp_path))
    assert_close(e.forward(b), want, oracle_scale(hd), what=".bin loader")
    nn = np.array([1, 2], np.int32)
    ne = np.array([0, 1], np.int32)
    tiny = gp.GraphBatch(nn, ne, np.zeros((3, 9), np.int32), np.array([[1, 0]], np.int32), np.zeros((1, 3), np.int32),
                         np.array([[0, .1, 0, 0], [0, -.2, 0, 0], [0, .3, 0, 0]], np.float32))
    want, hd = oracle.dgn_forward(tiny, [w2], dump_h=True)
    assert_close(e.forward(tiny), want, oracle_scale(hd), what="tiny graphs")
    tiny.node_eigen = None
    with pytest.raises(FlowGNNError) as ei:  # DGN needs the eigenvector column
        e.forward(tiny)
    assert ei.value.code == 1
    e.close()


def test_hep10k_size_properties(eng, oracle, w):
    """BASELINE config 5 size (10 000 hep10k-shaped graphs)."""
    b = gp.synth_hep10k_batch(10000, seed=1234)
    out = eng.forward(b)
    assert out.shape == (10000,) and np.isfinite(out).all()
    assert np.array_equal(out, eng.forward(b))
    # Batch split.  The default kernel for these dense tiles (dgn_layer_mfma_kernel) takes both aggregates as MFMAs with the tile's
    # adjacency: the order in which a row's 16 neighbour terms are summed then depends on where its graph sits in the tile, so a
    # different split gives the same values to fp32 rounding, NOT the same bits (stated tolerance, not bit identity) ...
    part = eng.forward(b.slice(4000, 4400))
    assert np.allclose(part, out[4000:4400], rtol=1e-5, atol=1e-5 * max(1.0, float(np.abs(out).max())))
    # ... while the in-edge walk (dgn_mfma_agg = 0) sums in CSR order and stays bit-identical under any split
    e2 = Engine("DGN", device=0, options={"dgn_mfma_agg": 0})
    e2.set_weights(w)
    walk = e2.forward(b)
    assert np.array_equal(e2.forward(b.slice(4000, 4400)), walk[4000:4400])
    e2.close()
    assert np.allclose(walk, out, rtol=1e-5, atol=1e-5 * max(1.0, float(np.abs(out).max())))
    # ... and ALL 10 000 graphs against the oracle
    want = oracle.dgn_forward(b, [w], nthreads=16)
    _, hd = oracle.dgn_forward(b.slice(0, 256), [w], dump_h=True, nthreads=16)  # the activation scale, from a slice
    assert_close(out, want, oracle_scale(hd), what="all 10 000 graphs")
    assert_close(walk, want, oracle_scale(hd), what="all 10 000 graphs, in-edge walk")


def test_split_range_fallback(oracle, w):
    """Same contract as GCN/GIN: dense200_res_relu_split_kernel raises the range flag beyond the f16 range and the engine
    repeats the pass on dgn_dense_kernel (fp32 MFMA)."""
    b = gp.synth_hep10k_batch(24, seed=47)
    e = Engine("DGN", device=0)
    e.set_weights(w)
    got, want = e.forward(b), oracle.dgn_forward(b, [w], nthreads=8)
    assert e.exact_reruns() == 0
    assert_close(got, want, what="in range")
    big = dict(w)
    k = "embedding_h_atom_embedding_list_weights"
    big[k] = w[k] * np.float32(1e6)
    e.set_weights(big)
    got, want = e.forward(b), oracle.dgn_forward(b, [big], nthreads=8)
    assert e.exact_reruns() == 1 and np.isfinite(got).all()
    assert_close(got, want, what="exact re-run")  # relative to the (huge) logits
    e.close()

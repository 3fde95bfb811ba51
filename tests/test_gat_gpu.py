"""GPU parity tests for GAT (BASELINE config 4): fused attention-gather + projection kernel vs the CPU oracle.
Tolerance (tests/parity.py): |gpu - oracle| <= 1e-4 * (scale + |oracle|), scale = the oracle's own largest activation (or logit where
no activations were dumped), measured per comparison (exp() of un-normalised scores on both sides)."""
import os

import numpy as np
import pytest

from flowgnn_amd import Engine, compute_graphs, graphpack as gp, weights
from tests.parity import assert_close, oracle_scale
from tests.test_oracle_gcn import directed_variant

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(__file__), "golden", "gat_molhiv48.npz")


@pytest.fixture(scope="module")
def w():
    return weights.synth_gat_weights(seed=7)


@pytest.fixture(scope="module")
def eng(w):
    e = Engine("GAT", device=0)
    e.set_weights(w)
    yield e
    e.close()


def test_forward_matches_oracle(eng, oracle, w):
    for b in (gp.synth_molhiv_batch(200, seed=31), directed_variant(gp.synth_molhiv_batch(40, seed=12)),
              gp.synth_hep10k_batch(12, seed=4, with_eigen=False)):
        got = eng.forward(b)
        want, hd = oracle.gat_forward(b, [w], dump_h=True, nthreads=8)
        scale = oracle_scale(hd)
        assert_close(eng.final_h(), hd[3], scale, what="o_3")
        assert_close(got, want, scale, what="logits")


def test_golden_vectors(eng):
    z = np.load(GOLDEN)
    b = gp.GraphBatch(z["nums_of_nodes"], z["nums_of_edges"], z["node_feature"], z["edge_list"], z["edge_attr"])
    assert_close(eng.forward(b), z["logits_synth_weights"], what="golden logits")


def test_reference_feature_quirk(oracle, w, monkeypatch):
    b = gp.synth_molhiv_batch(12, seed=3)
    monkeypatch.setenv("FLOWGNN_GAT_REFERENCE_QUIRK", "1")
    e = Engine("GAT", device=0)
    e.set_weights(w)
    assert_close(e.forward(b), oracle.gat_forward(b, [w], feature_offset_quirk=True), what="quirk")
    e.close()


def test_entry_point_bin_loader_and_edge_cases(tmp_path, oracle, w):
    b = gp.synth_molhiv_batch(9, seed=5)
    w2 = weights.synth_gat_weights(seed=8)
    rw = np.array([1, 0, 0, 0, 1, 0, 0, 0, 0], np.int32)
    want, hd = oracle.gat_forward(b, [w2], dump_h=True)
    assert_close(compute_graphs("GAT", b, [w, w2], rw), oracle.gat_forward(b, [w, w2], reload_weights=rw), oracle_scale(hd), what="two weight sets")
    weights.save_gat_weights(w2, str(tmp_path))
    e = Engine("GAT", device=0)
    e.load_weights_dir(str(tmp_path))
    assert_close(e.forward(b), want, oracle_scale(hd), what=".bin loader")
    nn = np.array([1, 2, 17], np.int32)
    ne = np.array([0, 1, 0], np.int32)
    nf = np.zeros((20, 9), np.int32)
    nf[:, 0] = np.arange(20) % 7
    tiny = gp.GraphBatch(nn, ne, nf, np.array([[1, 0]], np.int32), np.zeros((1, 3), np.int32))
    want, hd = oracle.gat_forward(tiny, [w2], dump_h=True)
    assert_close(e.forward(tiny), want, oracle_scale(hd), what="tiny graphs")
    e.close()


def test_full_molhiv_size_properties(eng, oracle, w):
    """BASELINE config 4 size (4 113 graphs)."""
    b = gp.synth_molhiv_batch(4113, seed=1234)
    out = eng.forward(b)
    assert out.shape == (4113,) and np.isfinite(out).all()
    assert np.array_equal(out, eng.forward(b))
    assert np.array_equal(eng.forward(b.slice(1000, 1500)), out[1000:1500])
    want, hd = oracle.gat_forward(b, [w], dump_h=True, nthreads=16)  # ALL 4 113 graphs
    assert_close(out, want, oracle_scale(hd), what="all 4 113 graphs")


def test_split_range_fallback(oracle, w):
    """Same contract as the other models: the split-f16 contractions of gat_layer_kernel raise the range flag when an
    operand leaves the f16 range and the engine repeats the pass on the fp32-MFMA variant.  The huge activations are made
    where nothing exponentiates them: layer 3's skip projection is scaled up and layer 4's linear projection is zeroed
    (all its scores are then 0)."""
    b = gp.synth_molhiv_batch(48, seed=53)
    e = Engine("GAT", device=0)
    try:
        e.set_weights(w)
        got, want = e.forward(b), oracle.gat_forward(b, [w], nthreads=8)
        assert e.exact_reruns() == 0
        assert_close(got, want, what="in range")
        big = {k: v.copy() for k, v in w.items()}
        big["skip_proj_weights"][3] *= np.float32(1e6)
        big["linear_proj_weights"][4] = 0.0
        e.set_weights(big)
        got, want = e.forward(b), oracle.gat_forward(b, [big], nthreads=8)
        assert np.isfinite(want).all() and np.abs(want).max() > 1e3
        assert e.exact_reruns() == 1 and np.isfinite(got).all()
        assert_close(got, want, what="exact re-run")  # relative to the (huge) logits
    finally:
        e.close()

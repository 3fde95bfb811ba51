"""GPU parity tests for GCN (BASELINE config 3): HIP path through the C ABI vs the CPU oracle.
Tolerance: |gpu - oracle| <= 1e-4 + 1e-4 |oracle| (fp32 both sides, MFMA fma chain vs scalar mul+add)."""
import os

import numpy as np
import pytest

from flowgnn_amd import Engine, FlowGNNError, GCN_compute_graphs, graphpack as gp, weights
from tests.test_oracle_gcn import directed_variant

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(__file__), "golden", "gcn_molpcba48.npz")


def close(a, b, rtol=1e-4, atol=1e-4):
    return np.allclose(a, b, rtol=rtol, atol=atol)


@pytest.fixture(scope="module")
def w():
    return weights.synth_gcn_weights(seed=7)


@pytest.fixture(scope="module")
def eng(w):
    e = Engine("GCN", device=0)
    e.set_weights(w)
    yield e
    e.close()


def test_forward_matches_oracle(eng, oracle, w):
    b = gp.concat_batches([gp.synth_molpcba_batch(200, seed=31), directed_variant(gp.synth_molpcba_batch(56, seed=32))])
    got = eng.forward(b)
    want, xd = oracle.gcn_forward(b, [w], dump_h=True, nthreads=8)
    assert np.isfinite(got).all()
    assert close(got, want), np.abs(got - want).max()
    assert close(eng.final_h(), xd[4], atol=2e-4), np.abs(eng.final_h() - xd[4]).max()


def test_golden_vectors(eng):
    z = np.load(GOLDEN)
    b = gp.GraphBatch(z["nums_of_nodes"], z["nums_of_edges"], z["node_feature"], z["edge_list"], z["edge_attr"])
    assert close(eng.forward(b), z["logits_synth_weights"])


def test_hep10k_shape(eng, oracle, w):
    b = gp.synth_hep10k_batch(16, seed=5, with_eigen=False)
    assert np.allclose(eng.forward(b), oracle.gcn_forward(b, [w], nthreads=8), rtol=2e-4, atol=5e-4)


def test_edge_cases_and_errors(eng, oracle, w):
    nn = np.array([1, 2, 33], np.int32)
    ne = np.array([0, 1, 0], np.int32)
    nf = np.zeros((36, 9), np.int32)
    nf[:, 0] = np.arange(36) % 119
    b = gp.GraphBatch(nn, ne, nf, np.array([[1, 0]], np.int32), np.array([[4, 5, 1]], np.int32))
    assert close(eng.forward(b), oracle.gcn_forward(b, [w]))
    bad = gp.synth_molpcba_batch(3, seed=1)
    bad.edge_attr[0, 1] = 6
    with pytest.raises(FlowGNNError) as ei:
        eng.forward(bad)
    assert ei.value.code == 3


def test_reference_entry_point_and_bin_loader(tmp_path, oracle, w):
    b = gp.synth_molpcba_batch(9, seed=5)
    w2 = weights.synth_gcn_weights(seed=8)
    rw = np.array([1, 0, 0, 0, 1, 0, 0, 0, 0], np.int32)
    got = GCN_compute_graphs(b, [w, w2], rw)
    assert close(got, oracle.gcn_forward(b, [w, w2], reload_weights=rw))
    weights.save_gcn_weights(w2, str(tmp_path))
    e = Engine("GCN", device=0)
    e.load_weights_dir(str(tmp_path))
    assert close(e.forward(b), oracle.gcn_forward(b, [w2]))
    e.close()


def test_full_molpcba_size_properties(eng, oracle, w):
    """BASELINE config 3 size (43 773 graphs): graph independence (permutation / sub-range give bit-identical
    logits), determinism, and an oracle check on a 128-graph sample."""
    b = gp.synth_molpcba_batch(43773, seed=1234)
    out = eng.forward(b)
    assert out.shape == (43773,) and np.isfinite(out).all()
    assert np.array_equal(out, eng.forward(b))
    assert np.array_equal(eng.forward(b.slice(20000, 21000)), out[20000:21000])
    rng = np.random.default_rng(0)
    idx = rng.choice(43773, 128, replace=False)
    sample = gp.concat_batches([b.slice(int(g), int(g) + 1) for g in idx])
    assert np.array_equal(eng.forward(sample), out[idx])
    assert close(out[idx], oracle.gcn_forward(sample, [w], nthreads=8))


def test_split_range_fallback(oracle, w):
    """The dense layers run as three f16 MFMAs per fp32 product (dense_split.h).  Reference-scale weights never
    need the exact-fp32 re-run; an embedding table scaled by 1e6 pushes activations past the f16 range, the
    engine repeats the pass on the fp32 kernels and still matches the oracle (relative: outputs are huge)."""
    b = gp.synth_molpcba_batch(150, seed=41)
    e = Engine("GCN", device=0)
    e.set_weights(w)
    assert close(e.forward(b), oracle.gcn_forward(b, [w], nthreads=8)) and e.exact_reruns() == 0
    big = dict(w)
    big["node_embedding_weight"] = w["node_embedding_weight"] * np.float32(1e6)
    e.set_weights(big)
    got, want = e.forward(b), oracle.gcn_forward(b, [big], nthreads=8)
    assert e.exact_reruns() == 1 and np.isfinite(got).all()
    assert np.allclose(got, want, rtol=1e-4, atol=1e-4 * np.abs(want).max()), np.abs(got - want).max()
    e.close()

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
    logits), determinism, and ALL 43 773 graphs against the oracle."""
    b = gp.synth_molpcba_batch(43773, seed=1234)
    out = eng.forward(b)
    assert out.shape == (43773,) and np.isfinite(out).all()
    assert np.array_equal(out, eng.forward(b))
    assert np.array_equal(eng.forward(b.slice(20000, 21000)), out[20000:21000])
    rng = np.random.default_rng(0)
    idx = rng.choice(43773, 128, replace=False)
    sample = gp.concat_batches([b.slice(int(g), int(g) + 1) for g in idx])
    assert np.array_equal(eng.forward(sample), out[idx])
    want = oracle.gcn_forward(b, [w], nthreads=16)
    assert close(out, want), np.abs(out - want).max()


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


def test_one_pass_front_end(oracle, w):
    """The default front end (gcn_tile_build_kernel + the resident kernel's own encoder: two launches, no index build, no x_0 rows in
    HBM) against the three-launch front end (index build, projected encoder, resident kernel): the same in-edge order (the
    descriptor's words are the CSR's, bit for bit), x_0 re-associated as (T01 + T234) + T5678 -- fp32 rounding only; bit-identical
    under a batch split; the CSR and the aggregation probe are still there on demand."""
    rng = np.random.default_rng(4)
    b = gp.concat_batches([gp.synth_molpcba_batch(700, seed=41), directed_variant(gp.synth_molpcba_batch(60, seed=42)),
                           gp.synth_molecule_batch(30, seed=43, mean_nodes=150.0, min_nodes=120, max_nodes=190)])
    el = b.edge_list.copy()
    eo = b.edge_offsets()
    for g in rng.integers(0, b.num_graphs, 40):  # duplicate edges and self loops
        if eo[g + 1] - eo[g] >= 2:
            el[eo[g]] = el[eo[g] + 1]
            el[eo[g + 1] - 1, 1] = el[eo[g + 1] - 1, 0]
    b = gp.GraphBatch(b.nums_of_nodes, b.nums_of_edges, b.node_feature, el, b.edge_attr)
    want = oracle.gcn_forward(b, [w], nthreads=8)
    res = {}
    for tb in (1, 0):
        e = Engine("GCN", device=0, options={"gcn_tile_build": tb})
        try:
            e.set_weights(w)
            e.profile_enable(True)
            res[tb] = e.forward(b).copy()
            names = set(e.profile_read())
            assert ("gcn_tile_build" in names) == (tb == 1) and ("build_csr" in names) == (tb == 0), names
            assert "gcn_resident" in names
            if tb == 1:
                assert np.array_equal(e.forward(b.slice(100, 600)), res[1][100:600])  # a different tiling: the same bits
                one = e.forward(b)
                row_ptr, src, eid, out_deg = e.csr()  # built on demand behind the one-pass run
                assert row_ptr[-1] == b.total_edges and out_deg.sum() == b.total_edges
                assert e.aggregation_only_ms(layer=0, iters=1) > 0.0
                assert np.array_equal(e.forward(b), one)
        finally:
            e.close()
    assert close(res[1], want) and close(res[0], want)
    assert np.allclose(res[1], res[0], rtol=2e-6, atol=2e-6), np.abs(res[1] - res[0]).max()

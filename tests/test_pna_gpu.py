"""GPU parity tests for PNA (BASELINE config 5): HIP path through the C ABI vs the CPU oracle.
Tolerance (tests/parity.py): |gpu - oracle| <= 1e-4 * (scale + |oracle|), scale = the oracle's own largest activation (or logit where
no activations were dumped), measured per comparison; std = sqrt(Q/n - mean^2) cancels in fp32 on both sides, which is why the
activation scale and not 1 is the unit."""
import os

import numpy as np
import pytest

from flowgnn_amd import Engine, compute_graphs, graphpack as gp, weights
from tests.parity import assert_close, oracle_scale
from tests.test_oracle_gcn import directed_variant

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(__file__), "golden", "pna_hep24.npz")


@pytest.fixture(scope="module")
def w():
    return weights.synth_pna_weights(seed=7)


@pytest.fixture(scope="module")
def eng(w):
    e = Engine("PNA", device=0)
    e.set_weights(w)
    yield e
    e.close()


def test_forward_matches_oracle(eng, oracle, w):
    for b in (gp.synth_hep10k_batch(48, seed=31, with_eigen=False), gp.synth_molhiv_batch(100, seed=3),
              directed_variant(gp.synth_molhiv_batch(40, seed=12))):
        got = eng.forward(b)
        want, hd = oracle.pna_forward(b, [w], dump_h=True, nthreads=8)
        scale = oracle_scale(hd)
        assert_close(eng.final_h(), hd[4], scale, what="h_4")
        assert_close(got, want, scale, what="logits")


def test_golden_vectors(eng):
    z = np.load(GOLDEN)
    b = gp.GraphBatch(z["nums_of_nodes"], z["nums_of_edges"], z["node_feature"], z["edge_list"], z["edge_attr"])
    assert_close(eng.forward(b), z["logits_synth_weights"], what="golden logits")  # relative to the logits themselves (order 1..10)


def test_entry_point_bin_loader_and_edge_cases(tmp_path, oracle, w):
    b = gp.synth_hep10k_batch(7, seed=5, with_eigen=False)
    w2 = weights.synth_pna_weights(seed=8)
    rw = np.array([1, 0, 0, 1, 0, 0, 0], np.int32)
    want, hd = oracle.pna_forward(b, [w2], dump_h=True)
    assert_close(compute_graphs("PNA", b, [w, w2], rw), oracle.pna_forward(b, [w, w2], reload_weights=rw), oracle_scale(hd), what="two weight sets")
    weights.save_pna_weights(w2, str(tmp_path))
    e = Engine("PNA", device=0)
    e.load_weights_dir(str(tmp_path))
    assert_close(e.forward(b), want, oracle_scale(hd), what=".bin loader")
    # single node without edges (sentinel min / max enter the arithmetic), two nodes one edge
    nn = np.array([1, 2], np.int32)
    ne = np.array([0, 1], np.int32)
    nf = np.zeros((3, 9), np.int32)
    tiny = gp.GraphBatch(nn, ne, nf, np.array([[1, 0]], np.int32), np.zeros((1, 3), np.int32))
    want, hd = oracle.pna_forward(tiny, [w2], dump_h=True)  # the sentinels (+-32) are activations here: the scale is theirs
    assert_close(e.forward(tiny), want, oracle_scale(hd, [32.0]), what="sentinel graphs")
    e.close()


def test_hep10k_size_properties(eng, oracle, w):
    """BASELINE config 5 size (10 000 hep10k-shaped graphs): graph independence and determinism at full size,
    oracle agreement on ALL 10 000 graphs."""
    b = gp.synth_hep10k_batch(10000, seed=1234, with_eigen=False)
    out = eng.forward(b)
    assert out.shape == (10000,) and np.isfinite(out).all()
    assert np.array_equal(out, eng.forward(b))
    assert np.array_equal(eng.forward(b.slice(4000, 4400)), out[4000:4400])
    want = oracle.pna_forward(b, [w], nthreads=16)
    _, hd = oracle.pna_forward(b.slice(0, 256), [w], dump_h=True, nthreads=16)  # the activation scale, from a slice (a full dump is 0.8 GB)
    assert_close(out, want, oracle_scale(hd), what="all 10 000 graphs")


def test_resident_kernel_is_the_per_layer_path_bit_for_bit(eng, w):
    """pna_resident_kernel (encoder + four layers + readout of a tile of whole graphs in one launch, h in LDS throughout) performs the
    operations of atom_encoder + 4 x pna_layer_fused + pool_mlp3 in the same order: the same bits, on full kNN tiles, on ragged
    molecule tiles (rows with 0..4 in-edges, the sentinel min / max) and on a batch that ends inside a tile."""
    per_layer = Engine("PNA", device=0, options={"pna_resident": 0})
    per_layer.set_weights(w)
    try:
        for b in (gp.synth_hep10k_batch(700, seed=77, with_eigen=False), gp.synth_molhiv_batch(3000, seed=78),
                  gp.synth_hep10k_batch(3, seed=79, with_eigen=False)):
            got, want = eng.forward(b), per_layer.forward(b)
            assert np.isfinite(got).all()
            assert np.array_equal(got, want), np.abs(got - want).max()
            # flowgnn_get_h after a resident run repeats the pass per layer and returns the same rows
            assert np.array_equal(eng.final_h(), per_layer.final_h())
    finally:
        per_layer.close()


def test_tile_build_from_the_edge_list_is_the_csr_path_bit_for_bit(eng, oracle, w):
    """pna_tile_build_kernel (descriptors straight from the caller's edge list: adjacency bit matrix per tile, no sort, no CSR in HBM)
    against launch_build_csr + pna_tile_desc_kernel: the same logits bit for bit -- with duplicate edges (up to 40 copies of one edge),
    self loops, rows of in-degree 0 and > 16, graphs of one node, and a tile that ends the batch."""
    rng = np.random.default_rng(5)
    hep = gp.synth_hep10k_batch(300, seed=81, with_eigen=False)
    eo = hep.edge_offsets()
    for g in range(0, 300, 7):  # duplicates and self loops inside every seventh graph
        e0, ne = int(eo[g]), int(eo[g + 1] - eo[g])
        for _ in range(12):
            i, k = rng.integers(0, ne, 2)
            hep.edge_list[e0 + i] = hep.edge_list[e0 + k]
        v = int(rng.integers(0, hep.nums_of_nodes[g]))
        hep.edge_list[e0 + int(rng.integers(0, ne))] = [v, v]
    g = 5  # forty copies of one edge: a row with 40+ in-edges, most of them the same source
    hep.edge_list[int(eo[g]):int(eo[g]) + 40] = hep.edge_list[int(eo[g])]
    mol = gp.synth_molhiv_batch(500, seed=82)
    one = gp.GraphBatch(np.array([1, 1, 2], np.int32), np.array([0, 1, 3], np.int32), np.zeros((4, 9), np.int32),
                        np.array([[0, 0], [0, 1], [0, 1], [1, 1]], np.int32), np.zeros((4, 3), np.int32))
    from_csr = Engine("PNA", device=0, options={"pna_tile_build": 0})
    from_csr.set_weights(w)
    try:
        for b in (hep, mol, one, gp.concat_batches([one, hep.slice(0, 9), one])):
            got, want = eng.forward(b), from_csr.forward(b)
            assert np.isfinite(got).all()
            assert np.array_equal(got, want), np.abs(got - want).max()
        assert_close(eng.forward(hep), oracle.pna_forward(hep, [w], nthreads=8), oracle_scale(oracle.pna_forward(hep, [w], dump_h=True, nthreads=8)[1]))
    finally:
        from_csr.close()
    bad = gp.synth_hep10k_batch(4, seed=83, with_eigen=False)
    bad.edge_list[3] = [0, 1000]  # out of range: refused as by the index build
    from flowgnn_amd import FlowGNNError
    with pytest.raises(FlowGNNError):
        eng.forward(bad)


def test_bin_packed_tiles_are_the_batch_order_tiles_bit_for_bit(eng, w):
    """Option pna_binpack (default on): the resident kernel walks tiles that flowgnn_set_batch BIN-PACKED from the batch's graphs (best
    fit, largest first: 90 % -> 98 % full on hep10k-shaped graphs) instead of tiles cut in batch order; the tile build writes the
    descriptors and the encoder's row numbers in tile order.  A row's aggregates and a graph's pooling depend on the row / the graph
    alone: the same logits bit for bit."""
    off = Engine("PNA", device=0, options={"pna_binpack": 0})
    off.set_weights(w)
    try:
        one = gp.GraphBatch(np.array([1, 1, 2], np.int32), np.array([0, 1, 3], np.int32), np.zeros((4, 9), np.int32),
                            np.array([[0, 0], [0, 1], [0, 1], [1, 1]], np.int32), np.zeros((4, 3), np.int32))
        hep = gp.synth_hep10k_batch(1500, seed=91, with_eigen=False)
        for b in (hep, gp.synth_molhiv_batch(3000, seed=92), gp.concat_batches([one, hep.slice(0, 9), one]), hep.slice(3, 4)):
            got, want = eng.forward(b), off.forward(b)
            assert np.isfinite(got).all() and np.array_equal(got, want), np.abs(got - want).max()
    finally:
        off.close()


def test_split_range_fallback(oracle, w):
    """Same contract as GCN/GIN: pna_dense_split_kernel raises the range flag when an aggregate leaves the f16 range
    and the engine repeats the pass on pna_dense_kernel (fp32 MFMA)."""
    b = gp.synth_hep10k_batch(24, seed=43)
    e = Engine("PNA", device=0)
    e.set_weights(w)
    got, want = e.forward(b), oracle.pna_forward(b, [w], nthreads=8)
    assert e.exact_reruns() == 0
    assert_close(got, want, what="in range")
    big = dict(w)
    big["node_embedding_weight"] = w["node_embedding_weight"] * np.float32(1e6)
    e.set_weights(big)
    got, want = e.forward(b), oracle.pna_forward(b, [big], nthreads=8)
    assert e.exact_reruns() == 1 and np.isfinite(got).all()
    assert_close(got, want, what="exact re-run")  # relative to the (huge) logits
    e.close()

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


def _with_duplicates(hep, seed):
    """Duplicate edges and self loops inside every seventh graph; forty copies of one edge in graph 5."""
    rng = np.random.default_rng(seed)
    eo = hep.edge_offsets()
    for g in range(0, hep.num_graphs, 7):
        e0, ne = int(eo[g]), int(eo[g + 1] - eo[g])
        for _ in range(12):
            i, k = rng.integers(0, ne, 2)
            hep.edge_list[e0 + i] = hep.edge_list[e0 + k]
        v = int(rng.integers(0, hep.nums_of_nodes[g]))
        hep.edge_list[e0 + int(rng.integers(0, ne))] = [v, v]
    if hep.num_graphs > 5:
        hep.edge_list[int(eo[5]):int(eo[5]) + 40] = hep.edge_list[int(eo[5])]
    return hep


def _without_duplicates(b):
    """The batch with every repeated (u, v) of a graph dropped (first copy kept)."""
    eo = b.edge_offsets()
    keep = np.ones(b.total_edges, bool)
    ne = b.nums_of_edges.copy()
    for g in range(b.num_graphs):
        e = b.edge_list[eo[g]:eo[g + 1]].astype(np.int64)
        _, first = np.unique(e[:, 0] * 65536 + e[:, 1], return_index=True)
        k = np.zeros(len(e), bool)
        k[first] = True
        keep[eo[g]:eo[g + 1]] = k
        ne[g] = int(k.sum())
    return gp.GraphBatch(b.nums_of_nodes.copy(), ne, b.node_feature.copy(), b.edge_list[keep].copy(), b.edge_attr[keep].copy(),
                         None if b.node_eigen is None else b.node_eigen.copy())


def test_resident_kernel_is_the_per_layer_path_bit_for_bit(eng, w):
    """dgn_resident_kernel (records from the caller's arrays, then encoder + four layers + readout of a tile of whole graphs in ONE
    launch: h in registers and as split rows in LDS throughout, the weights streamed) performs the operations of atom_encoder +
    dgn_rowinfo + 4 x dgn_layer_mfma_kernel + dgn_pool_part_mlp3<false> (rows pooled in row order) in the same order on the same tiles:
    the same bits -- on full kNN tiles, on
    ragged molecule tiles (forced onto the matrix pipe), on a batch that ends inside a tile; flowgnn_get_h after a resident run repeats
    the pass per layer."""
    per_layer = Engine("DGN", device=0, options={"dgn_resident": 0, "dgn_fold_readout": 0, "dgn_mfma_agg": 1})
    forced = Engine("DGN", device=0, options={"dgn_resident": 2, "dgn_mfma_agg": 1, "dgn_binpack": 0})  # (the per-layer path's tiles: graphs in batch order)
    for e in (per_layer, forced):
        e.set_weights(w)
    try:
        mol = with_eigen(gp.synth_molhiv_batch(3000, seed=78), 1)  # (its repeated bonds: rows that repeat ONE source add the copy alike in both
        for b in (gp.synth_hep10k_batch(700, seed=77), _without_duplicates(mol), gp.synth_hep10k_batch(3, seed=79)):  # kernels; the rest is toleranced, below)
            want = per_layer.forward(b)
            assert np.isfinite(want).all()
            got = forced.forward(b)
            assert np.array_equal(got, want), np.abs(got - want).max()
            # the default engine bin-packs the graphs into fuller tiles (option dgn_binpack): a graph's place in its tile, and with it the
            # order of its matrix-pipe sums, changes -- the same values to fp32 rounding (the stated 1e-5 of a batch split)
            packed = eng.forward(b)
            assert np.allclose(packed, want, rtol=1e-5, atol=1e-5 * max(1.0, float(np.abs(want).max()))), np.abs(packed - want).max()
            assert np.array_equal(packed, eng.forward(b))
            assert np.array_equal(forced.final_h(), per_layer.final_h())
        forced.profile_enable(True)
        forced.forward(gp.synth_hep10k_batch(64, seed=80))
        prof_names = set(forced.profile_read())
        assert prof_names == {"dgn_tile_build", "dgn_resident"}, prof_names  # two launches per step
    finally:
        for e in (per_layer, forced):
            e.close()


def test_resident_kernel_duplicate_edges_and_edge_cases(eng, oracle, w):
    """Rows with duplicate in-edges (the adjacency mask carries no multiplicities) re-sum their in-edges from the caller's edge list:
    against the oracle and against the in-edge walk, with up to 40 copies of one edge, self loops, graphs of one node, rows without
    in-edges, and out-of-range edges / node features refused as by the index build."""
    hep = _with_duplicates(gp.synth_hep10k_batch(300, seed=81), 5)
    one = gp.GraphBatch(np.array([1, 1, 2], np.int32), np.array([0, 1, 3], np.int32), np.zeros((4, 9), np.int32),
                        np.array([[0, 0], [0, 1], [0, 1], [1, 1]], np.int32), np.zeros((4, 3), np.int32),
                        np.array([[0, .1, 0, 0], [0, -.2, 0, 0], [0, .3, 0, 0], [0, .05, 0, 0]], np.float32))
    forced = Engine("DGN", device=0, options={"dgn_resident": 2})
    walk = Engine("DGN", device=0, options={"dgn_resident": 0, "dgn_mfma_agg": 0})
    forced.set_weights(w)
    walk.set_weights(w)
    try:
        for b in (hep, one, gp.concat_batches([one, hep.slice(0, 9), one]), with_eigen(gp.synth_molhiv_batch(3000, seed=78), 1)):
            got = forced.forward(b)
            want, hd = oracle.dgn_forward(b, [w], dump_h=True, nthreads=8)
            assert_close(got, want, oracle_scale(hd), what="resident vs oracle")
            assert_close(got, walk.forward(b), oracle_scale(hd), what="resident vs in-edge walk")
        bad = gp.synth_hep10k_batch(4, seed=83)
        bad.edge_list[3] = [0, 1000]
        with pytest.raises(FlowGNNError):
            forced.forward(bad)
        bad = gp.synth_hep10k_batch(4, seed=84)
        bad.node_feature[5, 0] = 119
        with pytest.raises(FlowGNNError):
            forced.forward(bad)
    finally:
        forced.close()
        walk.close()


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

"""GPU parity tests for GIN: HIP path (through the C ABI) vs the CPU oracle.

Tolerance (SURVEY 8c / BASELINE.json north_star "stated float tolerance"):
    |gpu - oracle| <= 1e-4 + 1e-4 * |oracle|   on logits and per-node embeddings
(fp32 both sides; only the order inside dot products differs: MFMA fma chain, k permuted).
Index bookkeeping (CSR vs the reference's load_graph tables) is bit-exact.
"""
import os

import numpy as np
import pytest

from flowgnn_amd import Engine, FlowGNNError, GIN_compute_graphs, graphpack as gp, weights

pytestmark = pytest.mark.gpu

RTOL = 1e-4
ATOL = 1e-4
GOLDEN = os.path.join(os.path.dirname(__file__), "golden", "gin_molhiv64.npz")


def close(a, b):
    return np.allclose(a, b, rtol=RTOL, atol=ATOL)


@pytest.fixture(scope="module")
def eng(gin_weights):
    e = Engine("GIN", device=0)
    e.set_weights(gin_weights)
    yield e
    e.close()


def expected_csr_from_oracle(oracle, batch):
    """Per-destination ordered in-edge lists implied by the reference's per-PE tables."""
    no, eo = batch.node_offsets(), batch.edge_offsets()
    row_ptr = [0]
    src = []
    for g in range(batch.num_graphs):
        n = int(batch.nums_of_nodes[g])
        el = batch.edge_list[eo[g]:eo[g + 1]]
        t = oracle.gin_load_graph(el, batch.edge_attr[eo[g]:eo[g + 1]], n)
        per_dst = [[] for _ in range(n)]
        for pe in range(4):
            pos = 0
            for u in range(n):
                for _ in range(int(t["degree_tables"][pe][u])):
                    v = int(t["neighbor_tables"][pe][pos]) * 4 + pe
                    per_dst[v].append(u + int(no[g]))
                    pos += 1
        for v in range(n):
            src.extend(per_dst[v])
            row_ptr.append(len(src))
    return np.asarray(row_ptr, np.int32), np.asarray(src, np.int32)


def test_csr_bit_exact_vs_reference_tables(eng, oracle):
    b = gp.concat_batches([gp.synth_molhiv_batch(40, seed=21), gp.add_virtual_nodes(gp.synth_molhiv_batch(8, seed=22))])
    # duplicate edges and a self loop exercise the tie-break (input order)
    b.edge_list[3] = b.edge_list[1]
    b.edge_list[5] = [2, 2]
    eng.forward(b)
    row_ptr, src, eid, out_deg = eng.csr()
    want_rp, want_src = expected_csr_from_oracle(oracle, b)
    assert np.array_equal(row_ptr, want_rp)
    assert np.array_equal(src, want_src)
    ge = b.global_edges()
    assert np.array_equal(ge[eid, 0], src)                         # eid really is the input edge
    assert np.array_equal(np.sort(eid), np.arange(b.total_edges))  # a permutation
    assert np.array_equal(out_deg, np.bincount(ge[:, 0], minlength=b.total_nodes))
    # ties (same u, same v) keep input order
    v_of = np.repeat(np.arange(b.total_nodes), np.diff(row_ptr))
    key = v_of.astype(np.int64) * (1 << 40) + src.astype(np.int64) * (1 << 20)
    same = key[1:] == key[:-1]
    assert (eid[1:][same] > eid[:-1][same]).all()


@pytest.mark.parametrize("n,e", [(200, 900), (250, 2000), (500, 5500), (1500, 12000), (3000, 20000)])
def test_csr_size_classes(eng, n, e):
    """One graph per size class of the index build (per-graph LDS kernels of 256/1024, 256/2048, 512/6144 and
    2048/16384 nodes/edges, then the flat global path), between small graphs: same CSR as a stable sort by
    (destination, source, input index)."""
    rng = np.random.default_rng(n)
    small = gp.synth_molhiv_batch(3, seed=n)
    nn = np.array([n], np.int32)
    nf = np.zeros((n, 9), np.int32)
    nf[:, 0] = rng.integers(0, 119, n)
    el = rng.integers(0, n, (e, 2)).astype(np.int32)
    ea = np.stack([rng.integers(0, 5, e), rng.integers(0, 6, e), rng.integers(0, 2, e)], 1).astype(np.int32)
    big = gp.GraphBatch(nn, np.array([e], np.int32), nf, el, ea)
    b = gp.concat_batches([small, big, small])
    eng.forward(b)
    row_ptr, src, eid, out_deg = eng.csr()
    ge = b.global_edges()
    order = np.lexsort((np.arange(len(ge)), ge[:, 0], ge[:, 1]))
    assert np.array_equal(eid, order)
    assert np.array_equal(src, ge[order, 0])
    assert np.array_equal(row_ptr, np.concatenate([[0], np.cumsum(np.bincount(ge[:, 1], minlength=b.total_nodes))]))
    assert np.array_equal(out_deg, np.bincount(ge[:, 0], minlength=b.total_nodes))


def test_forward_matches_oracle(eng, oracle, gin_weights):
    b = gp.synth_molhiv_batch(256, seed=31)
    got = eng.forward(b)
    want, hd = oracle.gin_forward(b, [gin_weights], dump_h=True, nthreads=8)
    assert np.isfinite(got).all()
    assert close(got, want), np.abs(got - want).max()
    assert close(eng.final_h(), hd[5]), np.abs(eng.final_h() - hd[5]).max()


def test_golden_vectors(eng):
    z = np.load(GOLDEN)
    b = gp.GraphBatch(z["nums_of_nodes"], z["nums_of_edges"], z["node_feature"], z["edge_list"], z["edge_attr"])
    got = eng.forward(b)
    assert close(got, z["logits_synth_weights"]), np.abs(got - z["logits_synth_weights"]).max()
    n4 = int(b.nums_of_nodes[:4].sum())
    assert close(eng.final_h()[:n4], z["h_first4_graphs"][5])


def test_hep10k_shape_and_virtual_nodes(eng, oracle, gin_weights):
    b = gp.synth_hep10k_batch(24, seed=5, with_eigen=False)
    got = eng.forward(b)
    want = oracle.gin_forward(b, [gin_weights], nthreads=8)
    # degree-16 sum aggregation grows activations: relative tolerance carries it
    assert np.allclose(got, want, rtol=2e-4, atol=1e-3), np.abs(got - want).max()
    v = gp.add_virtual_nodes(gp.synth_molhiv_batch(32, seed=6))
    got = eng.forward(v)
    want = oracle.gin_forward(v, [gin_weights], nthreads=8)
    assert np.allclose(got, want, rtol=2e-4, atol=1e-3), np.abs(got - want).max()


def test_edge_cases(eng, oracle, gin_weights):
    # single node without edges; two nodes one edge; a graph whose node count is not a tile multiple
    nn = np.array([1, 2, 33], np.int32)
    ne = np.array([0, 1, 0], np.int32)
    nf = np.zeros((36, 9), np.int32)
    nf[:, 0] = np.arange(36) % 119
    b = gp.GraphBatch(nn, ne, nf, np.array([[1, 0]], np.int32), np.array([[4, 5, 1]], np.int32))
    got = eng.forward(b)
    want = oracle.gin_forward(b, [gin_weights])
    assert close(got, want), (got, want)
    # empty batch
    empty = gp.GraphBatch(np.zeros(0, np.int32), np.zeros(0, np.int32), np.zeros((0, 9), np.int32),
                          np.zeros((0, 2), np.int32), np.zeros((0, 3), np.int32))
    assert eng.forward(empty).shape == (0,)


def test_invalid_inputs_return_error_codes(gin_weights):
    e = Engine("GIN", device=0)
    e.set_weights(gin_weights)
    b = gp.synth_molhiv_batch(4, seed=1)
    bad = gp.GraphBatch(b.nums_of_nodes, b.nums_of_edges, b.node_feature, b.edge_list.copy(), b.edge_attr)
    bad.edge_list[0, 1] = 10_000
    with pytest.raises(FlowGNNError) as ei:
        e.forward(bad)
    assert ei.value.code == 2
    bad = gp.GraphBatch(b.nums_of_nodes, b.nums_of_edges, b.node_feature, b.edge_list, b.edge_attr.copy())
    bad.edge_attr[0, 0] = 5
    with pytest.raises(FlowGNNError) as ei:
        e.forward(bad)
    assert ei.value.code == 3
    bad = gp.GraphBatch(b.nums_of_nodes, b.nums_of_edges, b.node_feature.copy(), b.edge_list, b.edge_attr)
    bad.node_feature[0, 1] = 4
    with pytest.raises(FlowGNNError) as ei:
        e.forward(bad)
    assert ei.value.code == 4
    bad = gp.GraphBatch(b.nums_of_nodes.copy(), b.nums_of_edges, b.node_feature, b.edge_list, b.edge_attr)
    bad.nums_of_nodes[1] = 0
    with pytest.raises(FlowGNNError) as ei:
        e.set_batch(bad)
    assert ei.value.code == 1
    # the engine is still usable afterwards
    assert np.isfinite(e.forward(b)).all()
    e.close()
    e2 = Engine("GIN", device=0)
    with pytest.raises(FlowGNNError) as ei:  # no weights
        e2.forward(b)
    assert ei.value.code == 6
    e2.close()


def test_reference_entry_point_with_two_weight_sets(oracle, gin_weights):
    b = gp.synth_molhiv_batch(10, seed=5)
    w2 = weights.synth_gin_weights(seed=8)
    rw = np.array([1, 0, 0, 0, 1, 0, 0, 0, 0, 0], np.int32)
    got = GIN_compute_graphs(b, [gin_weights, w2], rw)
    want = oracle.gin_forward(b, [gin_weights, w2], reload_weights=rw)
    assert close(got, want), np.abs(got - want).max()


def test_weights_from_bin_directory(tmp_path, oracle, gin_weights):
    weights.save_gin_weights(gin_weights, str(tmp_path))
    e = Engine("GIN", device=0)
    e.load_weights_dir(str(tmp_path))
    b = gp.synth_molhiv_batch(16, seed=2)
    assert close(e.forward(b), oracle.gin_forward(b, [gin_weights]))
    with pytest.raises(FlowGNNError) as ei:
        e.load_weights_dir(str(tmp_path / "missing"))
    assert ei.value.code == 7
    e.close()


def test_full_molhiv_size_properties(eng, oracle, gin_weights):
    """BASELINE.json config 2 size (4 113 graphs): size-independent properties.
    (1) graphs are independent: permuting the graph order permutes the outputs bit-exactly;
    (2) running a sub-range alone gives bit-identical logits; (3) ALL 4 113 graphs match the oracle."""
    b = gp.synth_molhiv_batch(4113, seed=1234)
    out = eng.forward(b)
    assert out.shape == (4113,) and np.isfinite(out).all()
    assert np.array_equal(out, eng.forward(b))  # deterministic (no atomics in the float path)
    rng = np.random.default_rng(0)
    perm = rng.permutation(4113)
    shuffled = gp.concat_batches([b.slice(int(g), int(g) + 1) for g in perm])
    assert np.array_equal(eng.forward(shuffled), out[perm])
    assert np.array_equal(eng.forward(b.slice(1000, 1500)), out[1000:1500])
    want = oracle.gin_forward(b, [gin_weights], nthreads=16)
    assert close(out, want), np.abs(out - want).max()


def test_split_precision_and_range_fallback(oracle, gin_weights):
    """The default GIN layer kernel evaluates every fp32 product of the dense update as three f16 MFMAs
    (gin_split.hip).  (1) On reference-scale weights it agrees with the fp32 oracle far inside the stated
    tolerance and never needs the exact-fp32 re-run.  (2) Activations beyond the f16 range (here: the node
    embedding table scaled by 1e5) trip the range flag; the engine repeats the pass on the fp32 MFMA kernels and
    the result still matches the oracle (relative tolerance: the logits are ~1e5)."""
    b = gp.synth_molhiv_batch(200, seed=21)
    e = Engine("GIN", device=0)
    e.set_weights(gin_weights)
    got, want = e.forward(b), oracle.gin_forward(b, [gin_weights])
    assert np.abs(got - want).max() < 2e-5, np.abs(got - want).max()
    assert e.exact_reruns() == 0
    big = dict(gin_weights)
    big["node_embedding_weight"] = gin_weights["node_embedding_weight"] * np.float32(1e5)
    e.set_weights(big)
    got, want = e.forward(b), oracle.gin_forward(b, [big])
    assert e.exact_reruns() == 1
    assert np.isfinite(got).all()
    assert np.allclose(got, want, rtol=1e-4, atol=1e-4 * np.abs(want).max()), np.abs(got - want).max()
    e.run()  # same resident batch: it stays on the exact kernels, no second detour
    assert np.array_equal(e.results(), got) and e.exact_reruns() == 1
    e.close()

"""The reference's size caps -- 500 nodes and 5 500 edges per graph (MAX_NODE / MAX_EDGE, GIN/src/dcl.h:17-18) -- for all
six models: one graph at both caps (random endpoints, so with duplicate edges, self loops and rows of very different
in-degree) between ordinary molecules, against the oracle.  Sum aggregation over ~11 in-edges per node makes GIN's
activations grow by orders of magnitude per layer, so the comparison is relative to the largest activation the oracle saw
(and the split-f16 GIN path may hand the batch to the fp32 kernels: also covered)."""
import numpy as np
import pytest

from flowgnn_amd import Engine, graphpack as gp, weights

pytestmark = pytest.mark.gpu

MAX_NODE, MAX_EDGE = 500, 5500


def capped_batch(seed, eigen):
    rng = np.random.default_rng(seed)
    nf = np.stack([rng.integers(0, c, MAX_NODE) for c in (119, 4, 12, 12, 10, 6, 6, 2, 2)], 1).astype(np.int32)
    el = rng.integers(0, MAX_NODE, (MAX_EDGE, 2)).astype(np.int32)
    ea = np.stack([rng.integers(0, 5, MAX_EDGE), rng.integers(0, 6, MAX_EDGE), rng.integers(0, 2, MAX_EDGE)], 1).astype(np.int32)
    eig = None
    if eigen:
        eig = np.zeros((MAX_NODE, 4), np.float32)
        eig[:, 1] = rng.uniform(-1, 1, MAX_NODE)
    big = gp.GraphBatch(np.array([MAX_NODE], np.int32), np.array([MAX_EDGE], np.int32), nf, el, ea, eig)
    small = gp.synth_hep10k_batch(2, seed=seed, with_eigen=True) if eigen else gp.synth_molhiv_batch(3, seed=seed)
    if not eigen:
        small.node_eigen = None
    return gp.concat_batches([small, big, small])


@pytest.mark.parametrize("model", ["GIN", "GIN-VN", "GCN", "GAT", "PNA", "DGN"])
def test_graph_at_the_reference_caps(model, oracle):
    base = model.replace("-VN", "").lower()
    w = getattr(weights, f"synth_{base}_weights")(seed=7)
    b = capped_batch(41, eigen=(model == "DGN"))
    if model == "GIN-VN":  # the virtual node and its 2 N edges come on top of the caps, as in GIN-VN/src/host_load.cc:125-153
        b = gp.add_virtual_nodes(b)
    want, hd = getattr(oracle, f"{base}_forward")(b, [w], dump_h=True, nthreads=8)
    e = Engine(model, device=0)
    try:
        e.set_weights(w)
        got = e.forward(b)
    finally:
        e.close()
    assert np.isfinite(want).all() and np.isfinite(got).all()
    scale = max(1.0, float(np.abs(hd).max()))
    assert np.allclose(got, want, rtol=2e-4, atol=2e-4 * scale), (np.abs(got - want).max(), scale)


def star(leaves):
    """leaves -> hub (node 0): a graph beyond every per-graph size class of the index build when leaves > 16 384."""
    n = leaves + 1
    nf = np.zeros((n, 9), np.int32)
    nf[:, 0] = np.arange(n) % 7
    el = np.stack([np.arange(1, n), np.zeros(leaves, np.int64)], 1).astype(np.int32)
    ea = np.zeros((leaves, 3), np.int32)
    return gp.GraphBatch(np.array([n], np.int32), np.array([leaves], np.int32), nf, el, ea)


def test_flat_index_build_hub_rows_and_their_cap(oracle):
    """Graphs beyond the LDS classes take the flat index build, whose per-row rank sort is quadratic: a 3 000-leaf star next to
    a 2 100-node chain runs (checked against the oracle); a hub above MAX_FLAT_INDEGREE = 16 384 in-edges is refused with
    FLOWGNN_ERR_UNSUPPORTED instead of running for minutes (flowgnn_amd/csrc/common.h)."""
    from flowgnn_amd.engine import FlowGNNError
    w = weights.synth_gcn_weights(seed=7)
    e = Engine("GCN", device=0)
    try:
        e.set_weights(w)
        n = 2100
        chain = gp.GraphBatch(np.array([n], np.int32), np.array([n - 1], np.int32), np.zeros((n, 9), np.int32),
                              np.stack([np.arange(n - 1), np.arange(1, n)], 1).astype(np.int32), np.zeros((n - 1, 3), np.int32))
        b = gp.concat_batches([star(3000), chain])
        got = e.forward(b)
        want = oracle.gcn_forward(b, [w], nthreads=4)
        assert np.allclose(got, want, rtol=1e-4, atol=1e-4), np.abs(got - want).max()
        with pytest.raises(FlowGNNError) as ei:
            e.forward(star(17000))
        assert ei.value.code == 8, ei.value  # FLOWGNN_ERR_UNSUPPORTED
        got2 = e.forward(b)  # the engine is usable afterwards
        assert np.array_equal(got2, got)
    finally:
        e.close()

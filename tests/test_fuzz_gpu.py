"""Randomised batches against the oracle, all six models: graph sizes from 1 node to a few hundred, directed random edges
with duplicates and self loops, graphs without edges, isolated nodes, hub nodes -- the shapes no generator of "realistic"
molecules produces, in one batch, so tiles, waves and size classes meet every mix."""
import numpy as np
import pytest

from flowgnn_amd import Engine, graphpack as gp, weights

pytestmark = pytest.mark.gpu
CARD = (119, 4, 12, 12, 10, 6, 6, 2, 2)


def random_batch(seed, eigen):
    rng = np.random.default_rng(seed)
    G = int(rng.integers(20, 60))
    nn = rng.choice([1, 2, 3, 7, 16, 17, 31, 64, 65, 130, 260], size=G, p=[.1, .1, .1, .15, .1, .1, .1, .1, .05, .05, .05]).astype(np.int32)
    ne, el, nf = [], [], []
    for n in nn:
        kind = rng.integers(0, 5)
        if kind == 0 or n == 1 and kind < 3:
            e = 0
        elif kind == 1:
            e = int(n)                      # sparse
        elif kind == 2:
            e = int(min(4 * n, 1500))       # denser, many duplicates for small n
        else:
            e = int(rng.integers(1, 3 * n + 2))
        ed = rng.integers(0, n, (e, 2)).astype(np.int32)
        if kind == 3 and e > 4:             # a hub: many edges into node 0
            ed[: e // 2, 1] = 0
        ne.append(e); el.append(ed)
        nf.append(np.stack([rng.integers(0, c, n) for c in CARD], 1).astype(np.int32))
    E = int(sum(ne))
    ea = np.stack([rng.integers(0, 5, E), rng.integers(0, 6, E), rng.integers(0, 2, E)], 1).astype(np.int32)
    eig = None
    if eigen:
        eig = np.zeros((int(nn.sum()), 4), np.float32)
        eig[:, 1] = rng.uniform(-1, 1, int(nn.sum()))
    return gp.GraphBatch(nn, np.asarray(ne, np.int32), np.concatenate(nf), np.concatenate(el).reshape(-1, 2), ea, eig)


@pytest.mark.parametrize("model", ["GIN", "GIN-VN", "GCN", "GAT", "PNA", "DGN"])
def test_random_batches_match_the_oracle(model, oracle):
    base = model.replace("-VN", "").lower()
    w = getattr(weights, f"synth_{base}_weights")(seed=11)
    e = Engine(model, device=0)
    try:
        e.set_weights(w)
        for seed in range(6):
            b = random_batch(100 * seed + len(model), eigen=(model == "DGN"))
            if model == "GIN-VN":
                b = gp.add_virtual_nodes(b)
            want, hd = getattr(oracle, f"{base}_forward")(b, [w], dump_h=True, nthreads=8)
            got = e.forward(b)
            assert np.isfinite(want).all(), (model, seed)
            scale = max(1.0, float(np.abs(hd).max()))
            assert np.allclose(got, want, rtol=2e-4, atol=2e-4 * scale), (model, seed, np.abs(got - want).max(), scale)
            if model == "GIN":  # the index build is bit-exact: stable order by (destination, source, input index)
                row_ptr, src, eid, out_deg = e.csr()
                ge = b.global_edges()
                order = np.lexsort((np.arange(len(ge)), ge[:, 0], ge[:, 1]))
                assert np.array_equal(eid, order) and np.array_equal(src, ge[order, 0])
                assert np.array_equal(row_ptr, np.concatenate([[0], np.cumsum(np.bincount(ge[:, 1], minlength=b.total_nodes))]))
                assert np.array_equal(out_deg, np.bincount(ge[:, 0], minlength=b.total_nodes))
    finally:
        e.close()


def test_random_batches_q6_10_bit_exact(oracle):
    """The Q6.10 mode on the same kind of batches: integer patterns, so the GPU must equal the C oracle bit for bit."""
    w = weights.synth_gin_weights(seed=11)
    e = Engine("GIN", device=0)
    try:
        e.set_weights(w)
        e.set_numeric_mode("q6.10")
        for seed in range(4):
            b = random_batch(1000 + seed, eigen=False)
            got = e.forward(b)
            want_f, want_q = oracle.gin_forward_q(b, [w], nthreads=8)
            assert np.array_equal(got, want_f), (seed, np.abs(got - want_f).max())
            assert np.array_equal(np.rint(got.astype(np.float64) * 1024).astype(np.int64), want_q.astype(np.int64))
    finally:
        e.close()

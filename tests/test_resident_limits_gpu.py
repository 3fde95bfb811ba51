"""The graph-resident kernels (gin_resident_kernel, gat_resident_kernel, gcn_resident_kernel) own tiles of WHOLE graphs packed by
flowgnn_set_batch (GraphTiles: GIN / GAT <= 256 rows and 1 280 in-edges, GCN <= 192 / 960).  Covered here, against the oracle:
graphs exactly AT a tile limit (rows, edges) between ordinary molecules, many tiles with ragged fill, a batch with one graph just
beyond a limit (the whole batch then takes the per-layer kernels), and resident == per-layer results within the parity tolerance."""
import numpy as np
import pytest

from flowgnn_amd import Engine, graphpack as gp, weights

pytestmark = pytest.mark.gpu

LIMITS = {"GIN": (256, 1280), "GIN-VN": (256, 1280), "GAT": (256, 1280), "GCN": (192, 960)}
MODELS = ["GIN", "GIN-VN", "GAT", "GCN"]  # GIN-VN: the HUBS form of the resident kernel (rows of more than 8 in-edges are hub rows)


def random_graph(n, m, seed):
    rng = np.random.default_rng(seed)
    nf = np.stack([rng.integers(0, c, n) for c in (119, 4, 12, 12, 10, 6, 6, 2, 2)], 1).astype(np.int32)
    # a ring (every node has an in-edge and an out-edge) + random extra edges, duplicates and self loops included
    ring = np.stack([np.arange(n), (np.arange(n) + 1) % n], 1)
    extra = rng.integers(0, n, (m - n, 2))
    el = np.concatenate([ring, extra]).astype(np.int32)
    el = el[rng.permutation(m)]
    ea = np.stack([rng.integers(0, 5, m), rng.integers(0, 6, m), rng.integers(0, 2, m)], 1).astype(np.int32)
    return gp.GraphBatch(np.array([n], np.int32), np.array([m], np.int32), nf, el, ea)


def run(model, b, env=None, monkeypatch=None):
    if env:
        for k, v in env.items():
            monkeypatch.setenv(k, v)
    w = getattr(weights, f"synth_{model.replace('-VN', '').lower()}_weights")(seed=11)
    e = Engine(model, device=0)
    try:
        e.set_weights(w)
        return e.forward(b), w
    finally:
        e.close()
        if env:
            for k in env:
                monkeypatch.delenv(k)


def check(model, b, oracle, monkeypatch, env=None):
    got, w = run(model, b, env, monkeypatch)
    want, hd = getattr(oracle, f"{model.replace('-VN', '').lower()}_forward")(b, [w], dump_h=True, nthreads=8)
    scale = max(1.0, float(np.abs(hd).max()))
    assert np.isfinite(got).all()
    assert np.allclose(got, want, rtol=2e-4, atol=2e-4 * scale), (model, np.abs(got - want).max(), scale)
    return got


@pytest.mark.parametrize("model", MODELS)
def test_graphs_at_the_tile_limits(model, oracle, monkeypatch):
    rows, edges = LIMITS[model]
    mol = gp.synth_molhiv_batch(40, seed=3)
    at_rows = random_graph(rows, rows + 40, seed=1)          # fills a tile's rows alone
    at_edges = random_graph(edges // 8, edges, seed=2)        # fills a tile's edge budget with few rows (in-degree ~ 8)
    both = random_graph(rows, edges, seed=4)                  # both at once
    b = gp.concat_batches([mol.slice(0, 13), at_rows, mol.slice(13, 14), at_edges, both, mol.slice(14, 40)])
    resident = check(model, b, oracle, monkeypatch)
    per_layer = check(model, b, oracle, monkeypatch, env={f"FLOWGNN_{model.replace('-VN', '')}_RESIDENT": "0"})
    scale = max(1.0, float(np.abs(per_layer).max()))
    assert np.allclose(resident, per_layer, rtol=2e-4, atol=2e-4 * scale)
    # any split of the batch gives the same bits (tiles are re-packed, rows change lanes)
    r2, _ = run(model, b.slice(10, 20))
    assert np.array_equal(r2, resident[10:20])


@pytest.mark.parametrize("model", MODELS)
def test_one_graph_beyond_a_limit_sends_the_batch_to_the_per_layer_kernels(model, oracle, monkeypatch):
    rows, edges = LIMITS[model]
    mol = gp.synth_molhiv_batch(30, seed=8)
    for big in (random_graph(rows + 1, rows + 30, seed=5), random_graph(edges // 8, edges + 1, seed=6)):
        check(model, gp.concat_batches([mol.slice(0, 20), big, mol.slice(20, 30)]), oracle, monkeypatch)


@pytest.mark.parametrize("model", MODELS)
def test_many_ragged_tiles(model, oracle, monkeypatch):
    """2 000 molecules: ~200 tiles of different fill on a 256-workgroup grid, and 600 workgroup-strided tiles with 6 000."""
    for g, seed in ((2000, 21), (6000, 22)):
        b = gp.synth_molpcba_batch(g, seed=seed)
        if model == "GIN-VN":
            b = gp.add_virtual_nodes(b)
        got, w = run(model, b)
        want = getattr(oracle, f"{model.replace('-VN', '').lower()}_forward")(b, [w], nthreads=16)
        assert np.allclose(got, want, rtol=2e-4, atol=2e-4 * max(1.0, float(np.abs(want).max()))), np.abs(got - want).max()

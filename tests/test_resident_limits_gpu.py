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
    w = getattr(weights, f"synth_{model.replace('-VN', '').lower()}_weights")(seed=11)
    e = Engine(model, device=0, options=env or {})
    try:
        e.set_weights(w)
        return e.forward(b), w
    finally:
        e.close()


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
    per_layer = check(model, b, oracle, monkeypatch, env={f"{model.replace('-VN', '').lower()}_resident": 0})
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


def random_tileable_batch(seed, max_nodes=190, eigen=False):
    """tests/test_fuzz_gpu.py's shapes (1-node graphs, graphs without edges, isolated nodes, duplicates, self loops, hubs), with
    every graph inside the smallest tile (192 rows / 960 in-edges), so that the batch stays on the resident kernels."""
    rng = np.random.default_rng(seed)
    G = int(rng.integers(30, 120))
    nn = rng.choice([1, 2, 3, 7, 16, 17, 31, 64, 65, 130, 190], size=G, p=[.1, .1, .1, .15, .1, .1, .1, .1, .05, .05, .05]).astype(np.int32)
    nn = np.minimum(nn, max_nodes).astype(np.int32)
    ne, el, nf = [], [], []
    for n in nn:
        kind = rng.integers(0, 5)
        if kind == 0 or n == 1 and kind < 3:
            e = 0
        elif kind == 1:
            e = int(n)
        elif kind == 2:
            e = int(min(4 * n, 900))
        elif kind == 4 and n > 17:
            e = int(16 * n)  # kNN-like density (the fused PNA / DGN layers' fast path: every row of a wave has >= 16 in-edges) ...
        else:
            e = int(min(rng.integers(1, 3 * n + 2), 900))
        ed = rng.integers(0, n, (e, 2)).astype(np.int32)
        if kind == 3 and e > 4:  # a hub: half of the edges into node 0
            ed[: e // 2, 1] = 0
        if kind == 4 and n > 17:  # ... exactly 16 in-edges per node
            ed[:, 1] = np.repeat(np.arange(n), 16)
        ne.append(e); el.append(ed)
        nf.append(np.stack([rng.integers(0, c, n) for c in (119, 4, 12, 12, 10, 6, 6, 2, 2)], 1).astype(np.int32))
    E = int(sum(ne))
    ea = np.stack([rng.integers(0, 5, E), rng.integers(0, 6, E), rng.integers(0, 2, E)], 1).astype(np.int32)
    eig = None
    if eigen:
        eig = np.zeros((int(nn.sum()), 4), np.float32)
        eig[:, 1] = rng.uniform(-1, 1, int(nn.sum()))
    return gp.GraphBatch(nn, np.asarray(ne, np.int32), np.concatenate(nf), np.concatenate(el).reshape(-1, 2), ea, eig)


@pytest.mark.parametrize("model", MODELS + ["PNA", "DGN"])
def test_random_tileable_batches(model, oracle):
    import os
    seeds = range(int(os.environ.get("FLOWGNN_FUZZ_SEEDS", "8")))  # a soak run sets this to a few hundred
    base = model.replace("-VN", "").lower()
    w = getattr(weights, f"synth_{base}_weights")(seed=11)
    e = Engine(model, device=0)
    try:
        e.set_weights(w)
        for seed in seeds:
            # PNA: tiles of 256 rows / 4 608 in-edges, DGN: 128 / 2 560 (kind 4 graphs: 16 in-edges per node)
            b = random_tileable_batch(7000 + 13 * seed + len(model), max_nodes={"PNA": 190, "DGN": 128}.get(model, 190) if model in ("PNA", "DGN") else 190,
                                      eigen=(model == "DGN"))
            if model not in ("PNA", "DGN"):  # the molecule models' tiles hold at most 960 in-edges: no 16-per-node graphs of 65+ nodes
                keep = np.nonzero(b.nums_of_edges <= 900)[0]
                b = gp.concat_batches([b.slice(int(g), int(g) + 1) for g in keep])
            want, hd = getattr(oracle, f"{base}_forward")(b, [w], dump_h=True, nthreads=8)
            got = e.forward(b)
            scale = max(1.0, float(np.abs(hd).max()))
            assert np.isfinite(got).all(), (model, seed)
            assert np.allclose(got, want, rtol=2e-4, atol=2e-4 * scale), (model, seed, np.abs(got - want).max(), scale)
    finally:
        e.close()

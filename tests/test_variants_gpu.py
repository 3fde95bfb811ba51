"""Every kernel variant that carries a reported number, checked against the oracle.

The default forward of each model is covered by tests/test_<model>_gpu.py.  The engine also ships alternative kernels
behind named options (flowgnn_set_option; include/flowgnn.h, table in flowgnn_amd/csrc/engine.hip):
    gin_unfused=1        gin_aggregate_tiled_kernel + gin_mlp_kernel (the kernel behind `aggregation_roofline`)
    gin_agg_untiled=1    ... with the first, un-tiled aggregation kernel;  gin_agg_tile=64|256: other tilings
    gin_mfma=32 ("f32")         fp32-MFMA fused layer (the exact fallback)
    gin_split_nt=1|2     four-wave forms of the split-f16 layer kernel
    gin_head_fold=0      resident kernel with the last layer's second linear layer computed (readout not folded through it)
    gin_tile_build=1|0   graph-resident GIN with the one-pass front end (gin_tile_build_kernel + in-kernel encoder) / with the three-kernel one
    gin_resident=0       per-layer launches instead of the graph-resident multi-layer kernel (gat_resident=0 likewise)
    {gin,gat}_fold_readout=0   separate mean-pool + linear kernel
    gcn_unfused=1        tiled_aggregate_kernel<GcnAggPolicy> + dense100_split_kernel
    <m>_mfma=32         fp32 matrix pipe for every model
    csr_flat=1           global-memory index build
    hipgraph             covered by tests/test_hipgraph_gpu.py
Each variant gets a fresh Engine with the option set through the API and must match the CPU oracle to the same
tolerance as the default path.  The standalone aggregation kernels timed by flowgnn_run_aggregation_only are read back
through flowgnn_get_aggregate and compared with the message-passing equations evaluated on the rows they read
(GIN/src/message_passing.cc:136-145 + node_embedding.cc:117; GCN/src/message_passing.cc:158-167 +
GCN/src/node_embedding.cc:123-138).
"""
import numpy as np
import pytest

from flowgnn_amd import Engine, graphpack as gp, weights
from tests.numpy_ref import ED_OFF

pytestmark = pytest.mark.gpu


def fresh_forward(model, options, batch, w, want_h=False):
    e = Engine(model, device=0, options=options)  # flowgnn_set_option per entry, before weights and batch
    try:
        for k, v in options.items():
            assert e.get_option(k) == (32.0 if v == "f32" else float(v))
        e.set_weights(w)
        out = e.forward(batch)
        h = e.final_h() if want_h else None
        reruns = e.exact_reruns()
    finally:
        e.close()
    return out, h, reruns


def batch_for(model):
    if model in ("GIN", "GAT"):
        return gp.synth_molhiv_batch(300, seed=41)
    if model == "GIN-VN":
        return gp.add_virtual_nodes(gp.synth_molhiv_batch(200, seed=42))
    if model == "GCN":
        return gp.synth_molpcba_batch(300, seed=43)
    return gp.synth_hep10k_batch(40, seed=44, with_eigen=(model == "DGN"))


def oracle_fn(oracle, model):
    return {"GIN": oracle.gin_forward, "GIN-VN": oracle.gin_forward, "GCN": oracle.gcn_forward, "GAT": oracle.gat_forward,
            "PNA": oracle.pna_forward, "DGN": oracle.dgn_forward}[model]


# tolerance per model: as in the model's own parity test file
TOL = {"GIN": (1e-4, 1e-4), "GIN-VN": (2e-4, 1e-3), "GCN": (1e-4, 1e-4), "GAT": (2e-4, 2e-4), "PNA": (2e-4, 2e-3), "DGN": (2e-4, 2e-3)}

GIN_VARIANTS = [
    {"gin_tile_build": 1},   # one-pass front end: tile descriptors + encoder row numbers from the caller's arrays, h_0 computed by the tile loader
    {"gin_tile_build": 0},   # index build + atom encoder + tile prep as separate launches
    {"gin_unfused": 1},
    {"gin_unfused": 1, "gin_agg_untiled": 1},
    {"gin_unfused": 1, "gin_agg_tile": 64},
    {"gin_unfused": 1, "gin_agg_tile": 256},
    {"gin_mfma": "f32"},
    {"gin_head_fold": 0},
    {"gin_resident": 0},
    {"gin_resident": 0, "gin_split_nt": 1},
    {"gin_resident": 0, "gin_split_nt": 2},
    {"gin_resident": 0, "gin_fold_readout": 0},
    {"gin_fold_readout": 0},
    {"csr_flat": 1},
]


@pytest.mark.parametrize("env", GIN_VARIANTS, ids=lambda e: ",".join(f"{k}={v}" for k, v in e.items()))
@pytest.mark.parametrize("model", ["GIN", "GIN-VN"])
def test_gin_variants_match_oracle(oracle, model, env):
    b = batch_for(model)
    w = weights.synth_gin_weights(seed=7)
    got, h, reruns = fresh_forward(model, env, b, w, want_h=True)
    want, hd = oracle.gin_forward(b, [w], dump_h=True, nthreads=8)
    rtol, atol = TOL[model]
    assert np.isfinite(got).all()
    assert np.allclose(got, want, rtol=rtol, atol=atol), (env, np.abs(got - want).max())
    assert np.allclose(h, hd[5], rtol=rtol, atol=atol * 5), (env, np.abs(h - hd[5]).max())
    assert reruns == 0


OTHER_VARIANTS = [
    ("GCN", {"gcn_resident": 0}),
    ("GCN", {"gcn_unfused": 1}),
    ("GCN", {"gcn_mfma": "f32"}),
    ("GCN", {"gcn_unfused": 1, "gcn_mfma": "f32"}),
    ("GCN", {"csr_flat": 1}),
    ("GAT", {"gat_mfma": "f32"}),
    ("GAT", {"gat_resident": 0}),
    ("GAT", {"gat_fold_readout": 0}),
    ("GAT", {"gat_mfma": "f32", "gat_fold_readout": 0}),
    ("PNA", {"pna_mfma": "f32"}),
    ("PNA", {"pna_fused": 0}),
    ("DGN", {"dgn_mfma": "f32"}),
    ("DGN", {"dgn_fused": 0}),
    ("DGN", {"dgn_mfma_agg": 0}),   # fused layer with the in-edge walk (the default for sparse tiles)
    ("DGN", {"dgn_mfma_agg": 1}),   # ... with both aggregates as MFMAs over the tile's adjacency (the default for kNN-dense tiles)
    ("PNA", {"tile_nominal": 64, "tile_slack": 0}),
    ("DGN", {"tile_nominal": 128, "tile_slack": 0}),
]


@pytest.mark.parametrize("model,env", OTHER_VARIANTS, ids=lambda x: x if isinstance(x, str) else ",".join(f"{k}={v}" for k, v in x.items()))
def test_other_model_variants_match_oracle(oracle, model, env):
    b = batch_for(model)
    w = weights.SYNTH[model](seed=7)
    got, _, _ = fresh_forward(model, env, b, w)
    want = oracle_fn(oracle, model)(b, [w], nthreads=8)
    rtol, atol = TOL[model]
    scale = max(1.0, float(np.abs(want).max()))
    assert np.isfinite(got).all()
    assert np.allclose(got, want, rtol=rtol, atol=atol * scale), (model, env, np.abs(got - want).max())


# ---------------------------------------------------------------- standalone aggregation kernels (roofline probes)
def gin_aggregate_reference(h, batch, row_ptr, src, eid, eemb_l):
    """a[v] = h[v] + sum over in-edges in CSR order of relu(h[u] + ((0 + E[a0]) + E[5+a1]) + E[11+a2]), in float32 with
    the kernel's association: bit-exact target."""
    ea = batch.edge_attr.astype(np.int64)[eid]
    ee = np.zeros((len(eid), h.shape[1]), np.float32)
    for k in range(3):
        ee = ee + eemb_l[ea[:, k] + ED_OFF[k]]
    msg = np.maximum(h[src] + ee, np.float32(0))
    acc = np.zeros_like(h)
    deg = np.diff(row_ptr)
    for i in range(int(deg.max()) if len(deg) else 0):  # i-th in-edge of every row that has one: sequential per row
        rows = np.nonzero(deg > i)[0]
        acc[rows] = acc[rows] + msg[row_ptr[rows] + i]
    return acc + h


@pytest.mark.parametrize("env", [{}, {"gin_agg_untiled": 1}, {"gin_agg_tile": 64}, {"gin_agg_tile": 256}],
                         ids=["tiled128", "untiled", "tiled64", "tiled256"])
def test_gin_aggregation_probe_is_bit_exact(oracle, env):
    """gin_aggregate_tiled_kernel, the kernel `aggregation_roofline` in bench.py is measured on."""
    b = gp.concat_batches([gp.synth_molhiv_batch(500, seed=51), gp.add_virtual_nodes(gp.synth_molhiv_batch(20, seed=52)),
                           gp.synth_hep10k_batch(6, seed=53, with_eigen=False)])
    w = weights.synth_gin_weights(seed=7)
    e = Engine("GIN", device=0, options=env)
    e.set_weights(w)
    out = e.forward(b)
    row_ptr, src, eid, _ = e.csr()
    want_out, hd = oracle.gin_forward(b, [w], dump_h=True, nthreads=8)
    assert np.allclose(out, want_out, rtol=2e-4, atol=1e-3)
    for layer in (0, 3):
        h_in, agg = e.aggregate(layer)
        # what the probe read is a real layer input: h_0 (graph-resident path: only the encoder output is in HBM), h_4 (readout
        # folded into the last per-layer launch) or h_5
        dist = [float(np.abs(h_in - hd[k]).max()) / max(1.0, float(np.abs(hd[k]).max())) for k in range(6)]
        assert min(dist) < 1e-3, dist
        want = gin_aggregate_reference(h_in, b, row_ptr, src, eid, np.asarray(w["edge_embedding_weight"], np.float32)[layer])
        assert np.array_equal(agg, want), (layer, np.abs(agg - want).max())
    assert np.array_equal(e.forward(b), out)  # the probe leaves the engine usable
    e.close()


def test_gcn_aggregation_probe_matches_equations(oracle):
    """tiled_aggregate_kernel<GcnAggPolicy<true>> (the GCN unfused roofline probe): a = relu(BN_l(m + relu(x + root_l)/(deg+1)))."""
    b = gp.synth_molpcba_batch(400, seed=61)
    w = weights.synth_gcn_weights(seed=7)
    e = Engine("GCN", device=0)
    e.set_weights(w)
    out = e.forward(b)
    want_out, xd = oracle.gcn_forward(b, [w], dump_h=True, nthreads=8)
    assert np.allclose(out, want_out, rtol=1e-4, atol=1e-4)
    f64 = lambda a: np.asarray(a, np.float64)
    ge = b.global_edges()
    u, v = ge[:, 0], ge[:, 1]
    N = b.total_nodes
    outdeg = np.bincount(u, minlength=N).astype(np.float64)
    dinv = np.where(outdeg > 0, 1.0 / np.sqrt(outdeg + 1.0), 0.0)
    for layer in (0, 2):
        x, agg = e.aggregate(layer)
        # the rows it read: h[final_h] -- x_4 after the per-layer kernels, x_0 after the graph-resident kernel (which keeps x_1..x_4 on chip)
        assert any(np.abs(x - xd[k]).max() < 2e-4 * max(1.0, float(np.abs(xd[k]).max())) for k in (0, 4))
        ee = f64(w["edge_embedding_weight"])[layer][b.edge_attr.astype(np.int64) + ED_OFF[None, :]].sum(axis=1)
        m = np.zeros((N, 100))
        np.add.at(m, v, (dinv[u] * dinv[v])[:, None] * np.maximum(f64(x)[u] + ee, 0.0))
        pre = m + np.maximum(f64(x) + f64(w["convs_root_emb_weight"])[layer], 0.0) / (outdeg[:, None] + 1.0)
        bn = (pre - f64(w["bn_mean"])[layer]) / np.sqrt(f64(w["bn_var"])[layer] + 2.0 ** -10) * f64(w["bn_weight"])[layer] + f64(w["bn_bias"])[layer]
        want = np.maximum(bn, 0.0)
        assert np.allclose(agg, want, rtol=1e-4, atol=1e-4), (layer, np.abs(agg - want).max())
    e.close()


def test_entry_point_skips_identical_weight_reloads(oracle):
    """reload_weights = 1 on every graph with the SAME set (legal and cheap in the reference, GIN_compute.cc:51-63): same
    results as one load; different sets still switch."""
    from flowgnn_amd import GIN_compute_graphs
    b = gp.synth_molhiv_batch(6, seed=5)
    w1, w2 = weights.synth_gin_weights(seed=7), weights.synth_gin_weights(seed=8)
    rw = np.ones(6, np.int32)
    got = GIN_compute_graphs(b, [w1, w1, w1, w2, w2, w1], rw)
    want = oracle.gin_forward(b, [w1, w1, w1, w2, w2, w1], reload_weights=rw)
    assert np.allclose(got, want, rtol=1e-4, atol=1e-4), np.abs(got - want).max()


def test_options_api_and_fixed_point_aggregate_guard(oracle):
    """flowgnn_set_option: unknown keys are refused; an option invalidates the resident batch; the aggregation taps are refused
    (not run on stale or missing inputs) while a fixed-point mode is selected."""
    from flowgnn_amd import FlowGNNError
    b = gp.synth_hep10k_batch(8, seed=3, with_eigen=False)
    e = Engine("PNA", device=0)
    with pytest.raises(FlowGNNError) as ei:
        e.set_option("no_such_switch", 1)
    assert ei.value.code == 8
    for dev_only in ("pna_ablate", "gin_pingpong"):  # development hooks and the measured-slower ping-pong kernel: not in the shipped library
        with pytest.raises(FlowGNNError) as ei:
            e.set_option(dev_only, 1)
        assert ei.value.code == 8
    e.set_weights(weights.SYNTH["PNA"](seed=7))
    ref = e.forward(b)
    e.set_option("pna_fused", 0)
    with pytest.raises(FlowGNNError) as ei:
        e.run()  # the batch must be set again after an option change
    assert ei.value.code == 6
    two_kernel = e.forward(b)
    assert np.allclose(two_kernel, ref, rtol=2e-4, atol=2e-3)
    e.aggregate(0)  # float mode: available
    e.set_numeric_mode("q6.10")
    e.forward(b)
    with pytest.raises(FlowGNNError) as ei:
        e.aggregate(0)
    assert ei.value.code == 8
    with pytest.raises(FlowGNNError) as ei:
        e.aggregation_only_ms(0, 1)
    assert ei.value.code == 8
    e.close()
    for model, mk in (("GCN", gp.synth_molpcba_batch), ("DGN", lambda n, seed: gp.synth_hep10k_batch(n, seed=seed, with_eigen=True))):
        e = Engine(model, device=0)
        e.set_weights(weights.SYNTH[model](seed=7))
        e.forward(mk(4, seed=1))
        e.aggregate(0)
        e.set_numeric_mode("q6.10")
        e.forward(mk(40, seed=2))  # a LARGER batch: stale tiles of the float run must not be used
        with pytest.raises(FlowGNNError) as ei:
            e.aggregate(0)
        assert ei.value.code == 8
        e.set_numeric_mode("f32")
        out = e.forward(mk(40, seed=2))
        e.aggregate(0)
        assert np.isfinite(out).all()
        e.close()


def test_one_pass_front_end_is_the_same_index_build(oracle):
    """gin_tile_build_kernel orders a tile's edges exactly as build_csr + gin_tile_prep do (same descriptors), so the two front ends
    differ only by the encoder's re-associated table sum: logits agree to a few ulp, and each is bit-identical under batch splits
    (duplicates, self loops, a hub row and kNN-dense graphs included); validation errors are reported the same way."""
    from flowgnn_amd import FlowGNNError
    rng = np.random.default_rng(5)
    b = gp.concat_batches([gp.synth_molhiv_batch(300, seed=51), gp.synth_hep10k_batch(3, seed=53, with_eigen=False), gp.synth_molhiv_batch(50, seed=54)])
    el = b.edge_list.copy()
    el[5] = el[4]                      # duplicate edge
    el[11, 1] = el[11, 0]              # self loop
    b = gp.GraphBatch(b.nums_of_nodes, b.nums_of_edges, b.node_feature, el, b.edge_attr)
    w = weights.synth_gin_weights(seed=7)
    outs = {}
    for tb in (1, 0):
        e = Engine("GIN", device=0, options={"gin_tile_build": tb, "gin_resident_min_fill": 0})
        e.set_weights(w)
        outs[tb] = e.forward(b)
        row_ptr, src, eid, _ = e.csr()  # the tap builds the CSR on demand after a one-pass run
        assert row_ptr[-1] == b.total_edges and len(src) == b.total_edges
        assert np.array_equal(e.forward(b.slice(40, 300)), outs[tb][40:300])
        e.close()
    want = oracle.gin_forward(b, [w], nthreads=8)
    assert np.allclose(outs[1], want, rtol=1e-4, atol=1e-4)
    assert np.allclose(outs[1], outs[0], rtol=1e-5, atol=1e-5)  # (the kNN graphs' logits are large: relative, not absolute)
    bad = gp.GraphBatch(b.nums_of_nodes, b.nums_of_edges, b.node_feature, el.copy(), b.edge_attr.copy())
    bad.edge_list[7, 0] = 10 ** 6
    for tb in (1, 0):
        e = Engine("GIN", device=0, options={"gin_tile_build": tb})
        e.set_weights(w)
        with pytest.raises(FlowGNNError) as ei:
            e.forward(bad)
        assert ei.value.code == 2, ei.value
        e.close()
    bad2 = gp.GraphBatch(b.nums_of_nodes, b.nums_of_edges, b.node_feature.copy(), el, b.edge_attr)
    bad2.node_feature[3, 2] = 99
    e = Engine("GIN", device=0, options={"gin_tile_build": 1})
    e.set_weights(w)
    with pytest.raises(FlowGNNError) as ei:
        e.forward(bad2)
    assert ei.value.code == 4, ei.value
    e.close()


def test_pingpong_kernel_is_bit_identical_to_the_lock_step_one(oracle):
    """gin_pp_kernel (gin_pingpong=1: one half of the workgroup multiplies while the other gathers and loads; =2: the same with
    sixteen waves, eight per half and one column tile each) does exactly the
    arithmetic of gin_resident_kernel per row -- same bits -- on ragged half-tiles, on graphs beyond the half-tile limits (129..256
    nodes: routed to the eight-wave kernel, one tile each) and when one half runs out of half-tiles before the other."""
    from flowgnn_amd import FlowGNNError
    from tests.test_resident_limits_gpu import random_graph
    probe = Engine("GIN", device=0)
    try:
        probe.set_option("gin_pingpong", 1)
    except FlowGNNError:
        pytest.skip("gin_pp_kernel exists in development builds only (make DEV=1): measured slower, not shipped")
    finally:
        probe.close()
    w = weights.synth_gin_weights(seed=7)
    mol = gp.synth_molhiv_batch(700, seed=61)
    b = gp.concat_batches([mol.slice(0, 100), random_graph(128, 640, seed=1), random_graph(150, 330, seed=2), mol.slice(100, 101),
                           random_graph(256, 1280, seed=3), random_graph(100, 700, seed=4), mol.slice(101, 700)])
    outs = {}
    for pp in (2, 1, 0):
        e = Engine("GIN", device=0, options={"gin_pingpong": pp, "gin_tile_build": 0, "gin_resident_min_fill": 0})
        e.set_weights(w)
        outs[pp] = e.forward(b)
        for lo, hi in ((0, 1), (3, 4), (50, 53), (99, 106), (0, b.num_graphs)):  # one graph, odd numbers of half-tiles, the big graphs, the lot
            assert np.array_equal(e.forward(b.slice(lo, hi)), outs[pp][lo:hi]), (pp, lo, hi)
        assert e.exact_reruns() == 0
        e.close()
    assert np.array_equal(outs[1], outs[0])
    assert np.array_equal(outs[2], outs[0])
    want = oracle.gin_forward(b, [w], nthreads=8)
    assert np.allclose(outs[1], want, rtol=1e-4, atol=1e-4 * max(1.0, float(np.abs(want).max())))


def test_dgn_mfma_aggregation_on_awkward_tiles(oracle):
    """dgn_layer_mfma_kernel forced on (dgn_mfma_agg = 1) where its special cases live: duplicate edges (multiplicity > 1 is not a
    0 / 1 adjacency: the correction walk), self loops, rows without in-edges, rows with far more than 16 in-edges, sparse molecule
    tiles (most source blocks empty) and a graph that fills the 128-row tile."""
    from tests.test_resident_limits_gpu import random_graph
    rng = np.random.default_rng(3)
    hep = gp.synth_hep10k_batch(12, seed=7, with_eigen=True)
    el = hep.edge_list.copy()
    el[3] = el[2]; el[4] = el[2]      # a triple edge
    el[40, 1] = el[40, 0]             # a self loop
    hep = gp.GraphBatch(hep.nums_of_nodes, hep.nums_of_edges, hep.node_feature, el, hep.edge_attr, hep.node_eigen)
    def with_eig(b, seed):
        e = np.zeros((b.total_nodes, 4), np.float32)
        e[:, 1] = np.random.default_rng(seed).uniform(-1, 1, b.total_nodes)
        return gp.GraphBatch(b.nums_of_nodes, b.nums_of_edges, b.node_feature, b.edge_list, b.edge_attr, e)
    b = gp.concat_batches([hep, with_eig(gp.synth_molhiv_batch(30, seed=8), 1), with_eig(random_graph(128, 2560, seed=5), 2),
                           with_eig(random_graph(40, 900, seed=6), 3)])
    w = weights.SYNTH["DGN"](seed=7)
    want = oracle.dgn_forward(b, [w], nthreads=8)
    scale = max(1.0, float(np.abs(want).max()))
    outs = {}
    for mode in (1, 0):
        e = Engine("DGN", device=0, options={"dgn_mfma_agg": mode})
        e.set_weights(w)
        outs[mode] = e.forward(b)
        e.close()
        assert np.isfinite(outs[mode]).all()
        assert np.allclose(outs[mode], want, rtol=2e-4, atol=2e-3 * scale), (mode, np.abs(outs[mode] - want).max(), scale)
    assert np.allclose(outs[1], outs[0], rtol=1e-4, atol=1e-4 * scale)


def test_dgn_pooled_last_layer(oracle):
    """dgn_fold_readout: the last MFMA layer hands the readout per-wave partial sums instead of its rows.  Same logits (the mean over
    a graph's nodes is added in another order: tolerance), on kNN tiles, on tiles of many tiny graphs (several graphs per wave), on a
    graph that fills the tile; flowgnn_get_h afterwards repeats the pass with the rows kept and leaves the engine as it was."""
    from tests.test_resident_limits_gpu import random_graph

    def with_eig(b, seed):
        e = np.zeros((b.total_nodes, 4), np.float32)
        e[:, 1] = np.random.default_rng(seed).uniform(-1, 1, b.total_nodes)
        return gp.GraphBatch(b.nums_of_nodes, b.nums_of_edges, b.node_feature, b.edge_list, b.edge_attr, e)
    tiny = gp.concat_batches([with_eig(random_graph(n, 3 * n, seed=100 + n), n) for n in (1, 2, 3, 5, 1, 17, 2, 33, 4, 1, 1, 64, 7)])
    batches = [gp.synth_hep10k_batch(301, seed=12),
               gp.concat_batches([tiny, with_eig(gp.synth_molhiv_batch(120, seed=9), 4), with_eig(random_graph(128, 2560, seed=5), 2), tiny])]
    w = weights.SYNTH["DGN"](seed=7)
    for b in batches:
        want, hd = oracle.dgn_forward(b, [w], dump_h=True, nthreads=8)
        scale = max(1.0, float(np.abs(hd).max()))
        e1 = Engine("DGN", device=0, options={"dgn_mfma_agg": 1, "dgn_fold_readout": 1})
        e0 = Engine("DGN", device=0, options={"dgn_mfma_agg": 1, "dgn_fold_readout": 0})
        for e in (e1, e0):
            e.set_weights(w)
        o1, o0 = e1.forward(b), e0.forward(b)
        assert np.isfinite(o1).all()
        assert np.allclose(o1, want, rtol=2e-4, atol=2e-3 * scale), (np.abs(o1 - want).max(), scale)
        assert np.allclose(o1, o0, rtol=1e-5, atol=1e-5 * scale), np.abs(o1 - o0).max()
        h1, h0 = e1.final_h(), e0.final_h()  # e1: the pass is repeated with the rows kept (the same kernels as e0)
        assert np.array_equal(h1, h0)
        assert np.allclose(h1, hd[4], rtol=2e-4, atol=2e-3 * scale)
        assert np.array_equal(e1.forward(b), o1)  # and the pooled form is back afterwards
        e1.close(); e0.close()


def test_dgn_in_edge_pass_from_the_edge_list_is_the_csr_one():
    """dgn_rowinfo_direct: the matrix-pipe path's per-row pass (adjacency mask, wsum, abssum, duplicate count, out-degree) made from
    the caller's edge list, no index build.  Same bits as the pass over the CSR (sources ascending = set bits ascending), also with
    duplicate edges (the device then builds the CSR after all, for the layer kernels' correction walk), self loops and rows without
    in-edges; the CSR tap still answers afterwards."""
    from tests.test_resident_limits_gpu import random_graph
    hep = gp.synth_hep10k_batch(40, seed=7, with_eigen=True)
    el = hep.edge_list.copy()
    el[3] = el[2]; el[4] = el[2]; el[700] = el[699]  # a triple and a double edge
    el[40, 1] = el[40, 0]                              # a self loop
    dup = gp.GraphBatch(hep.nums_of_nodes, hep.nums_of_edges, hep.node_feature, el, hep.edge_attr, hep.node_eigen)

    def with_eig(b, seed):
        e = np.zeros((b.total_nodes, 4), np.float32)
        e[:, 1] = np.random.default_rng(seed).uniform(-1, 1, b.total_nodes)
        return gp.GraphBatch(b.nums_of_nodes, b.nums_of_edges, b.node_feature, b.edge_list, b.edge_attr, e)
    w = weights.SYNTH["DGN"](seed=7)
    for b in (gp.synth_hep10k_batch(257, seed=3), dup, gp.concat_batches([dup, with_eig(random_graph(128, 2560, seed=5), 2), with_eig(gp.synth_molhiv_batch(60, seed=8), 1)])):
        outs = {}
        for direct in (1, 0):
            e = Engine("DGN", device=0, options={"dgn_mfma_agg": 1, "dgn_rowinfo_direct": direct, "dgn_resident": 0})  # (the per-layer path: the resident kernel has its own tile build)
            e.set_weights(w)
            outs[direct] = e.forward(b)
            assert np.array_equal(e.forward(b), outs[direct])
            if direct:
                row_ptr, src, eid, out_deg = e.csr()  # built on demand
                assert row_ptr[-1] == b.total_edges and np.array_equal(np.bincount(b.global_edges()[:, 0], minlength=b.total_nodes), out_deg)
                assert np.array_equal(e.forward(b), outs[direct])
            e.close()
        assert np.array_equal(outs[1], outs[0])


@pytest.mark.parametrize("model", ["GIN", "GIN-VN", "GCN", "GAT", "PNA", "DGN"])
def test_invalid_inputs_are_reported_on_every_path(model):
    """An edge endpoint outside its graph, an edge attribute or a node feature outside its table: the reference indexes out of bounds
    (it never validates); here the run fails with the validation code -- on the default path (one-pass GIN, resident kernels, DGN's
    index pass from the edge list), on the per-layer kernels, and through the drop-in symbol (whatever range of the pipeline the bad
    graph lands in) -- and the engine works again with the next good batch."""
    from flowgnn_amd import FlowGNNError, compute_graphs
    base = model.replace("-VN", "").lower()
    w = getattr(weights, f"synth_{base}_weights")(seed=7)
    hep = model in ("PNA", "DGN")
    good = (gp.synth_hep10k_batch if hep else gp.synth_molhiv_batch)(300, seed=5)
    if model == "GIN-VN":
        good = gp.add_virtual_nodes(good)

    def broken(kind):
        el, ea, nf = good.edge_list.copy(), good.edge_attr.copy(), good.node_feature.copy()
        e_mid = int(good.edge_offsets()[150]) + 1
        if kind == "edge":
            el[e_mid, 0] = int(good.nums_of_nodes[150]) + 3  # source beyond its graph
        elif kind == "attr":
            ea[e_mid, 1] = 6
        else:
            nf[int(good.node_offsets()[150]) + 1, 2] = 12
        return gp.GraphBatch(good.nums_of_nodes, good.nums_of_edges, nf, el, ea, good.node_eigen)
    # (GAT takes the nine node features as NUMBERS, not as table rows: any integer is a valid input there)
    kinds = ["edge"] + (["feat"] if base != "gat" else []) + (["attr"] if base in ("gin", "gcn") else [])
    switch = {"GIN": {"gin_resident": 0}, "GIN-VN": {"gin_resident": 0}, "GCN": {"gcn_resident": 0}, "GAT": {"gat_resident": 0},
              "PNA": {"pna_fused": 0}, "DGN": {"dgn_resident": 0}}[model]
    for opts in ({}, switch):
        e = Engine(model, device=0, options=opts)
        e.set_weights(w)
        want = e.forward(good).copy()
        for kind in kinds:
            with pytest.raises(FlowGNNError) as ei:
                e.forward(broken(kind))
            assert ei.value.code in (2, 3, 4), (kind, ei.value.code)  # FLOWGNN_ERR_EDGE_RANGE / _EDGE_ATTR / _NODE_FEAT
            assert np.array_equal(e.forward(good), want)
        e.close()
    for kind in kinds:
        with pytest.raises(FlowGNNError) as ei:
            compute_graphs(model, broken(kind), [w])
        assert ei.value.code in (2, 3, 4)
    assert np.allclose(compute_graphs(model, good, [w]), want, rtol=1e-4, atol=1e-4 * max(1.0, float(np.abs(want).max())))


@pytest.mark.gpu
@pytest.mark.parametrize("model", ["GIN", "GCN", "GAT"])
def test_tiles_full_of_one_node_graphs(oracle, model):
    """A tile of the graph-resident kernels holds up to 256 (GCN: 192) WHOLE graphs when every graph is a single node without
    edges; the readout of a tile is done by a few lanes (GIN: the 64 lanes of one wave, looping), one graph each -- every one of
    700 such graphs, mixed with ordinary molecules, must come out, and equal to the oracle's."""
    rng = np.random.default_rng(11)
    nf = np.stack([rng.integers(0, c, 700) for c in (119, 4, 12, 12, 10, 6, 6, 2, 2)], 1).astype(np.int32)
    ones = gp.GraphBatch(np.ones(700, np.int32), np.zeros(700, np.int32), nf, np.zeros((0, 2), np.int32), np.zeros((0, 3), np.int32))
    mol = gp.synth_molhiv_batch(40, seed=5)
    b = gp.concat_batches([mol.slice(0, 20), ones, mol.slice(20, 40)])
    w = weights.SYNTH[model](seed=7)
    want = getattr(oracle, model.lower() + "_forward")(b, [w], nthreads=4)
    e = Engine(model, device=0)
    e.set_weights(w)
    got = e.forward(b)
    e.close()
    assert got.shape == want.shape
    assert np.allclose(got, want, rtol=1e-4, atol=1e-4), np.abs(got - want).max()


@pytest.mark.parametrize("model", ["GIN", "GCN", "PNA", "DGN"])
def test_packed_host_to_device_transfer_is_the_plain_one(model):
    """Option h2d_pack (default on for batches of >= 8 MB): flowgnn_set_batch narrows node_feature / edge_list / edge_attr on host
    threads (9 B per node, 5 B per edge), copies a quarter of the bytes from pinned memory and widens them on the GPU into the
    reference's int32 layout.  Same logits bit for bit as the plain copies -- through the engine and through the drop-in symbol --
    and values that do not fit the narrow types (a feature of 300 or -7, a node id of 70 000, an attribute of 9) are refused with
    the code the plain path gives."""
    from flowgnn_amd import FlowGNNError, compute_graphs, entry_set_option
    hep = model in ("PNA", "DGN")
    b = (gp.synth_hep10k_batch(2500, seed=11) if hep else gp.synth_molhiv_batch(9000, seed=11) if model == "GIN" else gp.synth_molpcba_batch(9000, seed=11))
    assert 4 * (b.node_feature.size + b.edge_list.size + (0 if hep else b.edge_attr.size)) >= 8 << 20
    w = weights.SYNTH[model](seed=7)
    outs = {}
    for pack in (16, 0):
        e = Engine(model, device=0, options={"h2d_pack": pack})
        e.set_weights(w)
        outs[pack] = e.forward(b)
        codes = []
        for kind in ("feat_big", "feat_neg", "node_big") + (() if hep else ("attr",)):
            nf, el, ea = b.node_feature.copy(), b.edge_list.copy(), b.edge_attr.copy()
            if kind == "feat_big":
                nf[1234, 3] = 300
            elif kind == "feat_neg":
                nf[77, 0] = -7
            elif kind == "node_big":
                el[4321, 1] = 70000
            else:
                ea[999, 1] = 9
            bad = gp.GraphBatch(b.nums_of_nodes, b.nums_of_edges, nf, el, ea, b.node_eigen)
            with pytest.raises(FlowGNNError) as ei:
                e.forward(bad)
            codes.append(ei.value.code)
        outs[("codes", pack)] = codes
        assert np.array_equal(e.forward(b), outs[pack])  # the engine works again with the next good batch
        e.close()
    assert np.array_equal(outs[16], outs[0])
    assert outs[("codes", 16)] == outs[("codes", 0)], (outs[("codes", 16)], outs[("codes", 0)])
    try:
        entry_set_option(model, "h2d_pack", 16)
        packed = compute_graphs(model, b, [w])
        entry_set_option(model, "h2d_pack", 0)
        plain = compute_graphs(model, b, [w])
    finally:
        entry_set_option(model, "h2d_pack", 16)
    assert np.array_equal(packed, plain)
    if model != "DGN":  # (DGN's matrix-pipe sums depend on the tile a graph lands in: the entry point cuts ranges)
        assert np.array_equal(packed, outs[16])


@pytest.mark.parametrize("model", ["GIN", "GIN-VN", "GCN"])
def test_bin_packed_tiles_are_the_batch_order_tiles_bit_for_bit(model):
    """Options gin_binpack / gcn_binpack (default on): the one-pass resident path walks tiles that flowgnn_set_batch BIN-PACKED from the batch's graphs
    (best fit, largest first, windows of 1 024 graphs: 95 % -> 99 % full, 4.6 % fewer tiles) instead of tiles cut in batch order.  A
    tile is then a list of graphs; the tile build writes everything the resident kernel reads in tile order, and a row's sums depend
    on the row alone: the same logits bit for bit -- ragged sizes, one-node graphs, graphs that fill a tile, a batch of one graph."""
    from tests.test_resident_limits_gpu import random_graph
    mol = gp.synth_molhiv_batch(5000, seed=91)
    one = gp.GraphBatch(np.array([1], np.int32), np.array([0], np.int32), np.zeros((1, 9), np.int32), np.zeros((0, 2), np.int32), np.zeros((0, 3), np.int32))
    big = random_graph(250 if model != "GCN" else 190, 600, seed=3)
    batches = [mol, gp.concat_batches([one, mol.slice(0, 40), big, one, one, mol.slice(40, 300), big]), one, mol.slice(7, 8)]
    if model == "GIN-VN":
        batches = [gp.add_virtual_nodes(b) for b in batches[:2]]
    w = weights.SYNTH[model](seed=7)
    on, off = Engine(model, device=0), Engine(model, device=0, options={"gin_binpack": 0, "gcn_binpack": 0})
    try:
        for e in (on, off):
            e.set_weights(w)
        for b in batches:
            got, want = on.forward(b), off.forward(b)
            assert np.isfinite(got).all() and np.array_equal(got, want), np.abs(got - want).max()
            assert np.array_equal(on.final_h(), off.final_h())
    finally:
        on.close()
        off.close()

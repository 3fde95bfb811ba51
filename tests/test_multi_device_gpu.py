"""Several engines behind one handle (flowgnn_create_multi; north_star: the batch dimension partitioned across the GPUs of a
node).  The 1-GPU box lists device 0 more than once -- two / three engines on one GPU, each with its own stream and host thread,
which an RCCL rank-per-GPU launch cannot do: the sharding (cut by sum(N + E)), the per-engine threads and the job-order result
assembly are exactly what runs on eight devices.  Graphs are independent, so the results must be BIT-identical to the
single-engine run, for every model, ragged shards, empty shards, weight-set runs that span a cut, and through the `host` binary."""
import os
import subprocess

import numpy as np
import pytest

from flowgnn_amd import Engine, EngineGroup, compute_graphs, entry_set_devices, graphpack as gp, shard_ranges_c, weights

pytestmark = pytest.mark.gpu


def devs(n):
    """n device ordinals for an engine group: device 0 listed n times on a one-GPU box (the default), or the devices of
    FLOWGNN_TEST_DEVICES=0,1,... taken round robin -- the same tests then run on several physical GPUs."""
    ids = [int(x) for x in os.environ.get("FLOWGNN_TEST_DEVICES", "0").split(",") if x.strip() != ""] or [0]
    return [ids[i % len(ids)] for i in range(n)]

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST = os.path.join(ROOT, "flowgnn_amd", "host")
MODELS = ["GIN", "GIN-VN", "GCN", "GAT", "PNA", "DGN"]


def batch_for(model, n=61):
    if model in ("GIN", "GAT"):
        return gp.synth_molhiv_batch(n, seed=41)
    if model == "GIN-VN":
        return gp.add_virtual_nodes(gp.synth_molhiv_batch(n, seed=42))
    if model == "GCN":
        return gp.synth_molpcba_batch(n, seed=43)
    return gp.synth_hep10k_batch(n // 2, seed=44, with_eigen=(model == "DGN"))  # sizes vary: cuts by work differ from cuts by count


# Bit identity under a batch split needs kernels that sum a row's terms in an order that depends on the row alone.  DGN's default
# kernel for kNN-dense tiles does not (its aggregation is an MFMA contraction over the tile's adjacency: the order depends on where
# the graph sits in its tile -- tests/test_dgn_gpu.py); the in-edge walk does, so the bit-identity tests select it for DGN.
OPTS = {"DGN": {"dgn_mfma_agg": 0}}


def single(model, b, w):
    e = Engine(model, device=0, options=OPTS.get(model, {}))
    try:
        e.set_weights(w)
        return e.forward(b)
    finally:
        e.close()


@pytest.mark.parametrize("devices", [devs(2), devs(3)], ids=["2x", "3x"])
@pytest.mark.parametrize("model", MODELS)
def test_group_is_bit_identical_to_one_engine(model, devices):
    b, w = batch_for(model), weights.SYNTH[model](seed=7)
    want = single(model, b, w)
    g = EngineGroup(model, devices, options=OPTS.get(model, {}))
    try:
        g.set_weights(w)
        got = g.forward(b)
        cuts = g.shards()
        assert cuts == shard_ranges_c(b.nums_of_nodes, b.nums_of_edges, len(devices))
        assert cuts[0][0] == 0 and cuts[-1][1] == b.num_graphs and all(a < c for a, c in cuts)  # ragged, non-empty
        assert np.array_equal(got, want)
        # the resident shards re-run (the timed loop of `host`), then a different batch on the same group
        g.run()
        assert np.array_equal(g.results(), want)
        # (shards stay large enough to take the same kernel path as the single engine: a shard whose graph tiles are under
        # half full goes to the per-layer kernels, whose sums associate differently -- same values to ~1e-7, not the same bits)
        b2 = b.slice(2, b.num_graphs - 1)
        assert np.array_equal(g.forward(b2), want[2:b.num_graphs - 1])
    finally:
        g.close()


def test_more_engines_than_graphs_and_empty_batch():
    b, w = gp.synth_molhiv_batch(2, seed=5), weights.synth_gin_weights(seed=7)
    want = single("GIN", b, w)
    g = EngineGroup("GIN", devs(4))
    try:
        g.set_weights(w)
        assert np.array_equal(g.forward(b), want)  # two shards are empty
        assert sum(1 for a, c in g.shards() if a == c) == 2
        assert g.forward(b.slice(0, 0)).shape == (0,)
    finally:
        g.close()


def test_group_options_num_tasks_and_fixed_point(oracle):
    b = gp.synth_molpcba_batch(40, seed=8)
    w = weights.synth_gin_weights(seed=7, num_tasks=6)
    g = EngineGroup("GIN", devs(2), options={"gin_resident": 0})
    try:
        g.set_num_tasks(6)
        g.set_weights(w)
        got = g.forward(b)
        want = oracle.gin_forward(b, [w], num_tasks=6, nthreads=8)
        assert got.shape == (40, 6) and np.allclose(got, want, rtol=1e-4, atol=1e-4)
        g.set_num_tasks(1)
        w1 = weights.synth_gin_weights(seed=7)
        g.set_weights(w1)
        g.set_numeric_mode("q6.10")
        _, want_q = oracle.gin_forward_q(b, [w1], nthreads=8)
        assert np.array_equal(np.round(g.forward(b) * 1024.0).astype(np.int64), want_q.astype(np.int64))
    finally:
        g.close()


@pytest.mark.parametrize("model", ["GIN", "DGN"])
def test_entry_points_on_two_engines_with_reloads_spanning_a_cut(model, oracle):
    """<M>_compute_graphs with flowgnn_entry_set_devices({0, 0}): every run of constant weight set is sharded on its own, so a
    reload in the middle of what would be one shard is honoured; same bits as the one-device entry point."""
    b = batch_for(model, 40)
    G = b.num_graphs
    w1, w2 = weights.SYNTH[model](seed=7), weights.SYNTH[model](seed=8)
    rw = np.zeros(G, np.int32)
    rw[0] = 1
    rw[G // 2 - 1] = 1  # just before the middle cut
    rw[G - 2] = 1       # a two-graph run at the end: one graph per engine
    sets = [w1, w2, w1]
    from flowgnn_amd import entry_set_option
    for k, v in OPTS.get(model, {}).items():
        entry_set_option(model, k, v)
    try:
        entry_set_devices([0])
        one = compute_graphs(model, b, sets, rw)
        entry_set_devices(devs(2))
        two = compute_graphs(model, b, sets, rw)
    finally:
        entry_set_devices([0])
        for k in OPTS.get(model, {}):
            entry_set_option(model, k, -1)
    assert np.array_equal(one, two)
    want = getattr(oracle, model.lower() + "_forward")(b, sets, reload_weights=rw, nthreads=8)
    assert np.allclose(two, want, rtol=2e-4, atol=2e-3), np.abs(two - want).max()


@pytest.mark.parametrize("model", ["GIN", "PNA"])
def test_host_binary_devices_flag(model, tmp_path, oracle):
    """`host <MODEL> --devices 0,0` (the C++ host of north_star on two engines): HLS_output.txt equals the one-device run's and
    matches the oracle."""
    w = weights.SYNTH[model](seed=7)
    batch = gp.synth_hep10k_batch(7, seed=3, with_eigen=False) if model == "PNA" else gp.synth_molhiv_batch(23, seed=3)
    gdir, wdir = tmp_path / "graphs", tmp_path / "weights"
    gp.write_pack(batch, str(gdir))
    weights.SAVERS[model](w, str(wdir))
    outs = []
    for devs in ("0", "0,0"):
        out = tmp_path / f"out_{devs.replace(',', '_')}.txt"
        r = subprocess.run([HOST, model, "--graphs", str(gdir), "--weights", str(wdir), "--trials", "2", "--out", str(out), "--devices", devs],
                           capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stdout + r.stderr
        outs.append(open(out).read())
    assert outs[0] == outs[1]
    got = np.array([float(ln.split(":")[1]) for ln in outs[1].strip().splitlines()], dtype=np.float32)
    want = getattr(oracle, model.lower() + "_forward")(batch, [w])
    assert np.allclose(got, want, rtol=3e-4, atol=3e-4 * max(1.0, np.abs(want).max()))


def test_dgn_default_kernel_on_two_engines_is_toleranced(oracle):
    """DGN's default (MFMA-aggregation) path under a split: same values to fp32 rounding, and right against the oracle."""
    b, w = batch_for("DGN"), weights.SYNTH["DGN"](seed=7)
    e = Engine("DGN", device=0)
    e.set_weights(w)
    one = e.forward(b)
    e.close()
    g = EngineGroup("DGN", devs(2))
    try:
        g.set_weights(w)
        two = g.forward(b)
    finally:
        g.close()
    want = oracle.dgn_forward(b, [w], nthreads=8)
    scale = max(1.0, float(np.abs(want).max()))
    assert np.allclose(one, two, rtol=1e-5, atol=1e-5 * scale)
    assert np.allclose(two, want, rtol=2e-4, atol=2e-3 * scale)


def test_group_compute_pipelines_ranges_and_matches_the_single_engine(oracle):
    """flowgnn_group_compute: the host batch cut into size x chunks ranges, engine i taking ranges i, i + size, ... (copy of one
    range under the kernels of another).  Same logits as one engine on the whole batch -- bit for bit on GIN (graphs are
    independent and every range stays on the resident kernel), in job order, for ragged cuts, for more ranges than graphs, and
    through the entry points with every pipeline setting."""
    from flowgnn_amd import EngineGroup, compute_graphs, entry_set_pipeline
    w = weights.synth_gin_weights(7)
    b = gp.synth_molhiv_batch(6000, seed=77)
    ref = Engine("GIN", device=0)
    ref.set_weights(w)
    want = ref.forward(b).copy()
    ref.close()
    for engines, chunks in ((1, 3), (2, 1), (2, 4), (3, 2)):
        grp = EngineGroup("GIN", devs(engines))
        grp.set_weights(w)
        got = grp.compute(b, chunks)
        assert np.array_equal(got, want), (engines, chunks, np.abs(got - want).max())
        small = b.slice(0, 3)  # fewer graphs than ranges: empty ranges are skipped; a one-graph range is one resident tile (the fill
        assert np.array_equal(grp.compute(small, chunks), want[:3])  # threshold does not count a batch's last tile): the same bits
        assert np.array_equal(grp.compute(b, chunks), want)  # engines re-used with other sizes
        grp.close()
    try:
        for setting in (0, 1, 2, 5):
            entry_set_pipeline(setting)
            assert np.array_equal(compute_graphs("GIN", b, [w]), want), setting
    finally:
        entry_set_pipeline(0)


def test_kernel_choice_follows_the_job_not_the_shard():
    """Choices the library makes by batch statistics are made from the JOB's totals (flowgnn_set_job_totals; groups and the ranges of
    flowgnn_group_compute hand them down), so a job runs on the same kernels at any device count.  DGN chooses its aggregation by
    density (matrix pipe for E >= 8 N): a job of 900 kNN-dense graphs followed by 300 sparse ones is dense as a whole, and
    its sparse shard must still take the matrix-pipe kernel -- alone, it takes the in-edge walk.  (The two agree to fp32 rounding; the
    point is that the CHOICE does not depend on the cut.)  GIN has no size-dependent choice left (the one-pass front end is the
    default at every size): a 66 000-graph job is bit-identical on one engine, on two, and through flowgnn_group_compute."""
    from flowgnn_amd.dist import shard_ranges
    dense = gp.synth_hep10k_batch(900, seed=5, k=16)
    sparse = gp.synth_hep10k_batch(300, seed=6, k=3)
    job = gp.concat_batches([dense, sparse])
    assert job.total_edges >= 8 * job.total_nodes and sparse.total_edges < 8 * sparse.total_nodes
    w = weights.SYNTH["DGN"](seed=7)

    def kernels_of(batch, totals):
        e = Engine("DGN", device=0)
        try:
            e.set_weights(w)
            if totals:
                e.set_job_totals(*totals)
            e.profile_enable(True)
            out = e.forward(batch)
            return out, set(e.profile_read())
        finally:
            e.close()

    want, whole = kernels_of(job, None)
    assert "dgn_resident" in whole  # the matrix-pipe path (graph-resident kernel behind dgn_tile_build)
    alone, k_alone = kernels_of(sparse, None)
    assert "dgn_resident" not in k_alone  # its own job: sparse, the in-edge walk
    shard, k_shard = kernels_of(sparse, (job.total_nodes, job.total_edges))
    assert "dgn_resident" in k_shard
    np.testing.assert_allclose(shard, want[900:], rtol=2e-5, atol=2e-5)
    np.testing.assert_allclose(alone, want[900:], rtol=2e-4, atol=2e-4)
    g = EngineGroup("DGN", devs(2))
    try:
        g.set_weights(w)
        for i in range(2):
            g.lib.flowgnn_profile_enable(g_member(g, i), 1)
        got = g.forward(job)
        np.testing.assert_allclose(got, want, rtol=2e-5, atol=2e-5)
        for i in range(2):
            assert "dgn_resident" in g_profile_names(g, i)
    finally:
        g.close()
    b, wg = gp.synth_molhiv_batch(66000, seed=77), weights.synth_gin_weights(seed=7)
    e = Engine("GIN", device=0)
    try:
        e.set_weights(wg)
        e.profile_enable(True)
        want = e.forward(b)
        assert "gin_tile_build" in e.profile_read()
    finally:
        e.close()
    g = EngineGroup("GIN", devs(2))
    try:
        g.set_weights(wg)
        assert np.array_equal(g.forward(b), want)
        assert np.array_equal(g.compute(b, 3), want)
    finally:
        g.close()


def g_member(group, i):
    import ctypes as C
    group.lib.flowgnn_group_engine.restype = C.c_void_p
    group.lib.flowgnn_group_engine.argtypes = [C.c_void_p, C.c_int]
    group.lib.flowgnn_profile_enable.argtypes = [C.c_void_p, C.c_int]
    return C.c_void_p(group.lib.flowgnn_group_engine(group._h, i))


def g_profile_names(group, i):
    """Kernel names member i of a group has launched since profiling was switched on for it (per-engine call on a group member)."""
    import ctypes as C
    lib = group.lib
    h = g_member(group, i)
    n = C.c_int(32)
    names = (C.c_char_p * 32)()
    ms = (C.c_double * 32)()
    cnt = (C.c_longlong * 32)()
    lib.flowgnn_profile_read.argtypes = [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_char_p), C.POINTER(C.c_double), C.POINTER(C.c_longlong)]
    assert lib.flowgnn_profile_read(h, C.byref(n), names, ms, cnt) == 0
    return {names[k].decode() for k in range(n.value) if cnt[k] > 0}


def test_group_state_after_compute_and_error_text():
    """flowgnn_group_compute leaves the engines on their last ranges: run / results / shards answer FLOWGNN_ERR_STATE (6) until the
    next set_batch instead of writing old shards at stale offsets; a successful call clears the error text of a failed one."""
    from flowgnn_amd.engine import FlowGNNError
    b, w = gp.synth_molhiv_batch(40, seed=3), weights.synth_gin_weights(seed=7)
    want = single("GIN", b, w)
    g = EngineGroup("GIN", devs(2))
    try:
        g.set_weights(w)
        assert np.array_equal(g.forward(b), want)
        small = b.slice(0, 7)
        assert np.array_equal(g.compute(small, 2), want[:7])
        for call in (g.run, g.results, g.shards):
            with pytest.raises(FlowGNNError) as ei:
                call()
            assert ei.value.code == 6
        assert b"flowgnn_group_set_batch" in g.lib.flowgnn_group_last_error(g._h)
        bad = b.slice(0, 5)
        bad.edge_list = bad.edge_list.copy()
        bad.edge_list[0, 0] = 10 ** 6
        with pytest.raises(FlowGNNError):
            g.forward(bad)
        assert b"engine" in g.lib.flowgnn_group_last_error(g._h)
        assert np.array_equal(g.forward(b), want)  # recovers, and ...
        g.lib.flowgnn_group_sync(g._h)
        assert b"engine" not in (g.lib.flowgnn_group_last_error(g._h) or b"")  # ... the old text is gone
    finally:
        g.close()


def test_shards_take_the_jobs_side_of_the_fill_threshold():
    """The graph-resident kernels are used when the batch's graph tiles pack >= 50 % full (flowgnn.h).  A job of LARGE graphs packs
    about that full, and a shard of it can pack to the other side: the group (and flowgnn_set_job_tile_fill for a multi-process
    caller) hands the JOB's fill down, so the shard runs the kernels the whole job would run -- the same bits (round-4 advisor)."""
    rng = np.random.default_rng(3)
    w = weights.synth_gin_weights(seed=7)
    found = None
    e = Engine("GIN", device=0)
    try:
        e.set_weights(w)
        for trial in range(200):
            # graphs of 108..150 nodes with six in-edges per node: one per tile (two exceed its 1 280 in-edges), fill = mean / 256 ~ 0.5
            b = gp.synth_molecule_batch(24, seed=1000 + trial, mean_nodes=float(rng.uniform(122, 134)), edges_per_node=6.0, min_nodes=108, max_nodes=150)
            f_job = e.graph_tile_fill(b.nums_of_nodes, b.nums_of_edges)
            cuts = shard_ranges_c(b.nums_of_nodes, b.nums_of_edges, 2)
            for (a, c) in cuts:
                f_sh = e.graph_tile_fill(b.nums_of_nodes[a:c], b.nums_of_edges[a:c])
                if f_sh < 0.5 <= f_job and f_job - f_sh > 0.01:  # (the other direction lands on the per-layer kernels, which agree to rounding only)
                    found = (b, a, c, f_job, f_sh)
                    break
            if found:
                break
        assert found, "no job above / shard below the threshold in 200 trials"
        b, a, c, f_job, f_sh = found
        e.profile_enable(True)
        want = e.forward(b)
        job_kernels = set(e.profile_read())
        assert ("gin_resident" in job_kernels) == (f_job >= 0.5)
    finally:
        e.close()
    sh = b.slice(a, c)
    # the shard as its own job: the other side of the threshold, other kernels (same values to fp32 rounding)
    e = Engine("GIN", device=0)
    try:
        e.set_weights(w)
        e.profile_enable(True)
        alone = e.forward(sh)
        assert ("gin_resident" in set(e.profile_read())) == (f_sh >= 0.5) != (f_job >= 0.5)
        np.testing.assert_allclose(alone, want[a:c], rtol=2e-4, atol=2e-4)
    finally:
        e.close()
    # told the job's fill (and totals): the job's kernels, the job's bits
    e = Engine("GIN", device=0)
    try:
        e.set_weights(w)
        e.set_job_totals(b.total_nodes, b.total_edges)
        e.set_job_tile_fill(f_job)
        e.profile_enable(True)
        told = e.forward(sh)
        assert ("gin_resident" in set(e.profile_read())) == (f_job >= 0.5)
        assert np.array_equal(told, want[a:c])
    finally:
        e.close()
    g = EngineGroup("GIN", devs(2))
    try:
        g.set_weights(w)
        assert np.array_equal(g.forward(b), want)
        assert np.array_equal(g.compute(b, 2), want)
    finally:
        g.close()

"""Differential fuzz: random batches (graph sizes 1..500, sparse to dense, duplicate edges, self loops, isolated nodes, tiny and huge
graphs mixed) through the HIP path vs the CPU oracle, plus consistency under a batch split and through the entry point.
usage: fuzz.py MODEL [seconds] [seed] [mode]      (dev tool; imports oracle/ like the tests do)
mode: f32 (default) | q (the fixed-point mode, BIT-exact against the Q oracle) | variants (random option sets against the default path)
      | entry (the drop-in symbol with two or three weight sets switched by reload_weights, NUM_TASK 1..5 where the model has it, every
        pipeline setting, against the oracle)
environment: FUZZ_WEIGHT_SCALE=x (layer weights scaled: range flags, exact re-runs), FUZZ_WEIGHTS=trained (the reference's trained set)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
import oracle
from flowgnn_amd import Engine, compute_graphs, graphpack as gp, weights

model = sys.argv[1] if len(sys.argv) > 1 else "GIN"
budget = float(sys.argv[2]) if len(sys.argv) > 2 else 60.0
seed0 = int(sys.argv[3]) if len(sys.argv) > 3 else 1
mode = sys.argv[4] if len(sys.argv) > 4 else "f32"
base = model.replace("-VN", "").lower()
w = getattr(weights, f"synth_{base}_weights")(seed=11)
if os.environ.get("FUZZ_WEIGHTS", "") == "trained":  # the reference's own trained set (a data fixture of the tests) instead of synthetic weights
    z = np.load(os.path.join(ROOT, "tests", "golden", f"ref_weights_{base}.npz"))
    w = {k: z[k] for k in z.files}
wscale = float(os.environ.get("FUZZ_WEIGHT_SCALE", "1"))  # > 1: activations grow layer by layer -- the f16 range flag and the exact-fp32 re-run get exercised
if wscale != 1.0:
    w = {k: (v * np.float32(wscale) if ("weight" in k.lower() or "mlp" in k.lower() or "conv" in k.lower()) and "emb" not in k.lower() and "bn" not in k.lower() else v) for k, v in w.items()}
ofn = getattr(oracle, f"{base}_forward")


def rand_graph(rng):
    kind = rng.integers(0, 6)
    n = int(rng.choice([1, 2, 3, 5, 17, 33, 64, 65, 100, 128, 129, 200, 256, 257, 400, 500])) if kind == 0 else int(rng.integers(1, 60))
    dens = rng.choice([0.0, 0.5, 1.0, 2.2, 4.0, 16.0])
    m = int(min(5500, max(0, round(n * dens + rng.integers(0, 3)))))
    if model in ("PNA", "DGN") and rng.random() < 0.5:
        m = int(min(5500, n * min(16, max(n - 1, 0))))
    el = rng.integers(0, n, (m, 2)).astype(np.int32)
    if m and rng.random() < 0.3:
        el[rng.integers(0, m, max(1, m // 10))] = el[rng.integers(0, m)]  # duplicates
    nf = np.stack([rng.integers(0, c, n) for c in (119, 4, 12, 12, 10, 6, 6, 2, 2)], 1).astype(np.int32)
    ea = np.stack([rng.integers(0, 5, m), rng.integers(0, 6, m), rng.integers(0, 2, m)], 1).astype(np.int32).reshape(m, 3)
    eig = None
    if model == "DGN":
        eig = np.zeros((n, 4), np.float32)
        eig[:, 1] = rng.uniform(-1, 1, n)
    return gp.GraphBatch(np.array([n], np.int32), np.array([m], np.int32), nf, el, ea, eig)


VARIANTS = {
    "GIN": [{"gin_tile_build": 1}, {"gin_tile_build": 0}, {"gin_resident": 0},
            {"gin_resident": 0, "gin_unfused": 1}, {"gin_fold_readout": 0}, {"gin_head_fold": 0}, {"gin_resident_min_fill": 0}, {"hipgraph": 0}, {"gin_binpack": 0}, {"tile_balance": 0}],
    "GIN-VN": [{"gin_tile_build": 1}, {"gin_tile_build": 0}, {"gin_resident": 0}, {"gin_resident": 0, "gin_unfused": 1}, {"gin_resident_min_fill": 0}, {"gin_binpack": 0}, {"tile_balance": 0}],
    "GCN": [{"gcn_resident": 0}, {"gcn_resident": 0, "gcn_unfused": 1}, {"hipgraph": 0}, {"gcn_tile_build": 0}, {"gcn_binpack": 0}, {"tile_balance": 0}],
    "GAT": [{"gat_resident": 0}, {"gat_fold_readout": 0}, {"hipgraph": 0}, {"tile_balance": 0}],
    "PNA": [{"pna_fused": 0}, {"pna_resident": 0}, {"pna_tile_build": 0}, {"hipgraph": 0}, {"pna_binpack": 0}, {"pna_binpack": 0, "pna_tile_build": 0}, {"tile_balance": 0}],
    "DGN": [{"dgn_fused": 0}, {"dgn_mfma_agg": 0}, {"dgn_resident": 2}, {"dgn_binpack": 0}, {"tile_balance": 0}, {"dgn_mfma_agg": 1, "dgn_resident": 0}, {"dgn_mfma_agg": 1, "dgn_resident": 0, "dgn_rowinfo_direct": 0}, {"dgn_mfma_agg": 1, "dgn_resident": 0, "dgn_fold_readout": 0}],
}[model]

e = Engine(model, 0)
e.set_weights(w)
if mode == "q":
    e.set_numeric_mode("q6.10")
t_end = time.time() + budget
it = 0
worst = 0.0
while time.time() < t_end:
    rng = np.random.default_rng(seed0 * 100003 + it)
    graphs = [rand_graph(rng) for _ in range(int(rng.integers(1, 40)))]
    if rng.random() < 0.3:
        mol = gp.synth_molhiv_batch(int(rng.integers(1, 300)), seed=int(rng.integers(1 << 30)))
        if model == "DGN":
            eg = np.zeros((mol.total_nodes, 4), np.float32); eg[:, 1] = rng.uniform(-1, 1, mol.total_nodes)
            mol = gp.GraphBatch(mol.nums_of_nodes, mol.nums_of_edges, mol.node_feature, mol.edge_list, mol.edge_attr, eg)
        graphs.insert(int(rng.integers(0, len(graphs) + 1)), mol)
    b = gp.concat_batches(graphs)
    if model == "GIN-VN":
        b = gp.add_virtual_nodes(b)
    if mode == "entry":
        from flowgnn_amd import entry_set_pipeline
        T = int(rng.integers(1, 6)) if base in ("gin", "gcn") else 1
        mk = getattr(weights, f"synth_{base}_weights")
        sets = [mk(seed=11 + k, num_tasks=T) if base in ("gin", "gcn") else mk(seed=11 + k) for k in range(int(rng.integers(1, 4)))]
        rw = np.zeros(b.num_graphs, np.int32)
        rw[0] = 1
        for k in range(1, len(sets)):
            if b.num_graphs > 1:
                rw[int(rng.integers(1, b.num_graphs))] = 1
        rw[0] = 1
        sets = sets[: int(rw.sum())]
        entry_set_pipeline(int(rng.choice([0, 1, 2, 3])))
        got = compute_graphs(model, b, sets, rw, **({"num_tasks": T} if T > 1 else {}))
        kw = {"num_tasks": T} if base in ("gin", "gcn") else {}
        want = ofn(b, sets, rw, nthreads=8, **kw)
        scale = max(1.0, float(np.abs(want).max()))
        ok = got.shape == want.shape and np.isfinite(got).all() and np.allclose(got, want, rtol=2e-4, atol=2e-3 * scale if model in ("PNA", "DGN", "GAT") else 2e-4 * scale)
        worst = max(worst, float(np.abs(got - want).max() / scale) if got.size else 0.0)
        if not ok:
            print(f"FAIL {model} mode entry seed {seed0} iter {it}: graphs {b.num_graphs} sets {len(sets)} tasks {T} rw {np.nonzero(rw)[0]} max|d| {np.abs(got - want).max():.3e} scale {scale:.3e}", flush=True)
            sys.exit(1)
        it += 1
        continue
    got = e.forward(b)
    if mode == "q":
        want = oracle.gin_forward_q(b, [w], nthreads=8) if base == "gin" else oracle.q_forward(model, b, [w], nthreads=8)[0]
        want = want[0] if isinstance(want, tuple) else want
        ok = np.array_equal(got, want)
        cut = int(rng.integers(0, b.num_graphs + 1))
        parts = [e.forward(b.slice(a, c)) for a, c in ((0, cut), (cut, b.num_graphs)) if c > a]
        ok_split = np.array_equal(np.concatenate(parts) if parts else got[:0], got)
        ok_ent, scale, err = True, 1.0, float(np.abs(got - want).max()) if got.size else 0.0
    else:
        want, hd = ofn(b, [w], dump_h=True, nthreads=8)
        if not np.isfinite(want).all() or not np.isfinite(hd).all():
            it += 1  # the reference arithmetic itself overflowed on this batch: nothing to compare
            continue
        scale = max(1.0, float(np.abs(hd).max()), float(np.abs(want).max()))  # (scaled head weights: the logits can be larger than any activation)
        # The fuzzer's own bounds sit a decade above the worst error ever seen per model (a wrong-head bug in a GAT experiment stayed
        # inside the tests' documented tolerance at 600x the usual error); scaled weights fall back to the documented ones.
        tight = {"GIN": 2e-5, "GIN-VN": 2e-5, "GCN": 2e-5, "GAT": 2e-6, "PNA": 1e-3, "DGN": 1e-5}[model]
        if wscale != 1.0:
            tight = 2e-3 if model in ("PNA", "DGN", "GAT") else 2e-4
        ok = np.isfinite(got).all() and np.allclose(got, want, rtol=tight, atol=tight * scale)
        cut = int(rng.integers(0, b.num_graphs + 1))
        parts = [e.forward(b.slice(a, c)) for a, c in ((0, cut), (cut, b.num_graphs)) if c > a]
        split = np.concatenate(parts) if parts else got[:0]
        # (PNA: a shard may pack below the fused kernel's fill threshold and run the two-kernel layers, whose std = sqrt(Q/n - mean^2)
        # rounds differently -- and that difference of squares amplifies roundings when the variance is small against the mean)
        ok_split = np.allclose(split, got, rtol=1e-4, atol=(5e-4 if model == "PNA" else 1e-4) * scale)
        if mode == "variants":
            opts = VARIANTS[int(rng.integers(0, len(VARIANTS)))]
            e2 = Engine(model, 0, options=opts)
            e2.set_weights(w)
            ent = e2.forward(b)
            e2.close()
        else:
            opts = "entry point"
            ent = compute_graphs(model, b, [w])
        ok_ent = np.isfinite(ent).all() and np.allclose(ent, got, rtol=1e-4, atol=(5e-4 if model == "PNA" else 1e-4) * scale)
        err = float(np.abs(got - want).max() / scale) if got.size else 0.0
    worst = max(worst, err)
    if not (ok and ok_split and ok_ent):
        print(f"FAIL {model} mode {mode} seed {seed0} iter {it}: graphs {b.num_graphs} nodes {b.total_nodes} edges {b.total_edges} oracle_ok {ok} split_ok {ok_split} "
              f"other_ok {ok_ent} ({opts if mode != 'q' else ''}) max|d| {np.abs(got - want).max():.3e} scale {scale:.3e} exact_reruns {e.exact_reruns()}", flush=True)
        sys.exit(1)
    it += 1
print(f"{model} [{mode}]: {it} random batches ok, worst |gpu - oracle| / scale = {worst:.2e}, exact_reruns {e.exact_reruns()}", flush=True)

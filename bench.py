#!/usr/bin/env python3
"""bench.py -- graphs/s of the GIN (dim 100) hot path on molhiv-shaped synthetic graphs.

    python bench.py --gpus N --steps K --warmup W [--model M] [--graphs G] [--scaling weak|strong]

A "step" = one full batched forward of the resident batch through the C ABI (flowgnn_run):
batched load_graph (CSR build) -> atom encoder -> 5 x (aggregation + node MLP) -> mean-pool +
head, for G graphs per GPU (weak scaling, default) or G graphs in the whole job cut by sum(N+E)
(strong scaling), followed for N > 1 by the RCCL all-gather that concatenates the per-graph results.
Inputs are resident in HBM when the timed region starts (the reference also times kernel execution
only: run_experiments.sh:44).  Rank 0 prints ONE JSON line, which carries `roofline`, `cpu_baseline`
(N = 1) and `parity` (GPU logits of the timed batch vs the oracle); the default single-GPU GIN run also measures the other
BASELINE configs after the timed region and reports them as `configs`.  A failed parity check -- headline or any config --
is exit code 3 after the line is printed.

Run plainly with --gpus N > 1 it starts the N ranks itself (one process per GPU, 127.0.0.1
rendezvous); under torchrun it uses the launcher's RANK / LOCAL_RANK / WORLD_SIZE.  It refuses to
run when the rank count or the visible GPU count differs from --gpus.

The default batch is the "roofline batch" of SURVEY 8d: 2^18 graphs per GPU, so that every
[N_tot][100] fp32 tensor (2.7 GB) is far larger than the 256 MiB Infinity Cache.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0       # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
FP32_MFMA_PEAK_TF = 157.3   # MI355X_MICROARCH.md: v_mfma_f32_16x16x4_f32 / 32x32x2_f32, dense
F16_MFMA_PEAK_TF = 2500.0   # MI355X_MICROARCH.md: dense f16/bf16 MFMA peak (measured here: 1947, tools/mfma_f16_split.hip)
FPGA_U50_GRAPHS_PER_S = 20214.0  # GIN molhiv on Alveo U50 (other hardware; informational)


def GIN_RESIDENT_BYTES(n, e, tiles=None):  # what gin_resident_kernel reads from HBM per launch (see the GIN entry below)
    return n * 4 + (tiles if tiles else n // 256 + 1) * 3584


# rows of a graph tile of each model's resident kernel: with the engine's tile fill of the batch (flowgnn_graph_tile_fill) that gives the
# number of tiles the batch really packs to -- what the per-tile descriptors in the `moved_bytes` formulas are counted by
TILE_ROWS = {"GIN": 256, "GIN-VN": 256, "GCN": 192, "PNA": 256, "DGN": 128}


# Per-model bench table.  Algorithmic work per launch of the two kernel classes (DESIGN.md "Kernels"):
#   aggregation bytes (unpadded): read h once + write the aggregates + edge index (8 B) + edge payload
#   transform flops: 2 x MACs of the dense update per node per layer
MODELS = {
    "GIN": dict(metric="graphs/sec on ogbg-molhiv (GIN, dim=100)", dataset="molhiv", graphs=1 << 18,
                agg_bytes=lambda n, e: n * 400 * 2 + e * 20, flops=lambda n, e: n * 80000,
                # fused layer: read h once + write h' + CSR (row_ptr 4 B/node, src 4 B + edge code 1 B per edge)
                # graph-resident kernel (all five layers in one launch) behind the one-pass front end (the default since round 4): per node
                # it reads the 4-byte encoder row numbers gin_tile_build wrote, per 256-row tile the 3 584-byte descriptor (CSR slice as
                # 16-bit words, row offsets); the encoder / edge tables and the weight stream are L2-resident; it writes 4 B per graph;
                # its bound is the f16 matrix pipe (5 layers x 3 x 80 000 flop per node)
                fused_bytes={"gin_layer_fused": lambda n, e: n * 400 * 2 + n * 4 + e * 5, "gin_resident": GIN_RESIDENT_BYTES},
                moved_bytes={"gin_resident": lambda n, e, t: GIN_RESIDENT_BYTES(n, e, t)},
                layers_per_launch={"gin_resident": 5}, mfma_bound_kernels=("gin_resident",),
                # dense layers' worth of products one launch EXECUTES: the single-task readout is folded through the last layer's
                # second linear layer (never computed), so 4.5 of the 5 layers count
                dense_layers_per_launch={"gin_resident": 4.5},
                hbm_kernels=("gin_aggregate",), mfma_kernels=("gin_resident", "gin_layer_fused", "gin_mlp"),
                workload="GIN dim=100, batched ogbg-molhiv-shaped graphs on MI355X (BASELINE configs[1])"),
    "GIN-VN": dict(metric="graphs/sec on ogbg-molhiv (GIN-VN, dim=100)", dataset="molhiv-vn", graphs=1 << 18,
                   agg_bytes=lambda n, e: n * 400 * 2 + e * 20, flops=lambda n, e: n * 80000,
                   fused_bytes={"gin_layer_fused": lambda n, e: n * 400 * 2 + n * 4 + e * 5, "gin_resident": GIN_RESIDENT_BYTES},
                   moved_bytes={"gin_resident": lambda n, e, t: GIN_RESIDENT_BYTES(n, e, t)},
                   layers_per_launch={"gin_resident": 5}, mfma_bound_kernels=("gin_resident",), dense_layers_per_launch={"gin_resident": 4.5},
                   hbm_kernels=("gin_aggregate",), mfma_kernels=("gin_resident", "gin_layer_fused", "gin_mlp"),
                   workload="GIN-VN dim=100 (virtual node per graph), ogbg-molhiv-shaped graphs"),
    "GCN": dict(metric="graphs/sec on ogbg-molpcba (GCN, dim=100)", dataset="molpcba", graphs=1 << 18,
                agg_bytes=lambda n, e: n * 400 * 2 + e * 24, flops=lambda n, e: n * 20000,
                # fused layer (aggregate + BN/root/degree epilogue + dense): rows in and out, CSR entry + norm + code per
                # edge, row bounds + out-degree per node; unfused dense layer: read a row, write a row
                # graph-resident kernel (five aggregations + four dense layers in one launch): priced on 5 aggregations' + 4 dense
                # layers' per-layer figures; behind the one-pass front end (round 5) it really moves one 3 584-byte descriptor per
                # 192-row tile (CSR slice, out-degrees, encoder row numbers; roofline.hbm_bytes_moved) -- the projected encoder table
                # and the weight stream are L2-resident
                fused_bytes={"gcn_layer_fused": lambda n, e: n * 400 * 2 + n * 8 + e * 9, "gcn_dense": lambda n, e: n * 400 * 2,
                             "gcn_resident": lambda n, e: 4 * (n * 400 * 2 + n * 8 + e * 9) + (n * 400 + n * 8 + e * 9)},
                moved_bytes={"gcn_resident": lambda n, e, t: t * 3584},
                layers_per_launch={"gcn_resident": 4},
                hbm_kernels=("gcn_aggregate",), mfma_kernels=("gcn_resident", "gcn_layer_fused", "gcn_dense"),
                workload="GCN dim=100, batched ogbg-molpcba-shaped graphs on MI355X (BASELINE configs[2])"),
    "GAT": dict(metric="graphs/sec on ogbg-molhiv (GAT, 4 heads x 16)", dataset="molhiv", graphs=1 << 18,
                agg_bytes=lambda n, e: n * (256 + 32) + n * 256 + (e + n) * 8, flops=lambda n, e: n * 16384,
                # the graph-resident kernel runs all five layers in one launch: priced on five times the per-layer figure (what it
                # really moves is 36 B of features per node + the CSR: roofline.hbm_bytes_moved)
                hbm_kernels=("gat_resident", "gat_layer"), mfma_kernels=(), layers_per_launch={"gat_resident": 5},
                moved_bytes={"gat_resident": lambda n, e, t: n * (36 + 4) + e * 4},
                workload="GAT 5-layer, 4 heads x 16, ogbg-molhiv-shaped graphs on MI355X (BASELINE configs[3])"),
    "PNA": dict(metric="graphs/sec on hep10k (PNA, dim=80)", dataset="hep10k", graphs=1 << 16,
                agg_bytes=lambda n, e: n * 320 + n * 320 * 4 + e * 8, flops=lambda n, e: n * 153600,
                # unfused split dense: read 4 aggregates + h, write h'; fused layer: read h (tile rows) + CSR, write h'
                # graph-resident kernel (encoder + four layers + readout in one launch): four layers of products; what it moves is 36 B of
                # node features + 4 B of out-degree per node and one 5 632-byte CSR descriptor per 256-row tile
                fused_bytes={"pna_dense": lambda n, e: n * (1280 + 320 + 320), "pna_layer_fused": lambda n, e: n * (320 + 320 + 4) + e * 4,
                             "pna_resident": lambda n, e: 4 * (n * (320 + 320 + 4) + e * 4)},
                moved_bytes={"pna_resident": lambda n, e, t: n * 36 + t * 6144},
                layers_per_launch={"pna_resident": 4},
                mfma_bound_kernels=("pna_layer_fused", "pna_resident"),
                hbm_kernels=("pna_aggregate",), mfma_kernels=("pna_resident", "pna_layer_fused", "pna_dense"),
                workload="PNA dim=80, hep10k-shaped kNN graphs on MI355X (BASELINE configs[4])"),
    "DGN": dict(metric="graphs/sec on hep10k (DGN, dim=100)", dataset="hep10k", graphs=1 << 16,
                agg_bytes=lambda n, e: n * 400 * 3 + e * 12, flops=lambda n, e: n * 40000,
                # unfused split dense: read both aggregates + h, write h'; fused layer: read h (tile rows) + CSR + eigenvector, write h'
                # graph-resident kernel (encoder + four layers + readout in one launch): four layers of products; what it moves is the
                # 48-byte record per row that dgn_tile_build writes (adjacency mask, wsum, abssum, degrees, eig1, encoder rows)
                fused_bytes={"dgn_dense": lambda n, e: n * (800 + 400 + 400), "dgn_layer_fused": lambda n, e: n * (400 + 300 + 44),  # rows in, rows out (average of 4 launches: the last one writes none, dgn_fold_readout), 32 B of stored in-edge pass + eigenvector entry + out-degree per row
                             "dgn_resident": lambda n, e: 4 * n * (400 + 400 + 44)},
                moved_bytes={"dgn_resident": lambda n, e, t: t * 6144},
                layers_per_launch={"dgn_resident": 4},
                mfma_bound_kernels=("dgn_resident",),
                hbm_kernels=("dgn_aggregate",), mfma_kernels=("dgn_resident", "dgn_layer_fused", "dgn_dense"),
                workload="DGN dim=100, hep10k-shaped kNN graphs on MI355X (BASELINE configs[4])"),
}


def make_batch(dataset: str, graphs: int, seed: int):
    from flowgnn_amd import graphpack as gp
    if dataset == "molhiv":
        return gp.synth_molhiv_batch(graphs, seed=seed)
    if dataset == "molhiv-vn":
        return gp.add_virtual_nodes(gp.synth_molhiv_batch(graphs, seed=seed))
    if dataset == "molpcba":
        return gp.synth_molpcba_batch(graphs, seed=seed)
    if dataset == "hep10k":
        return gp.synth_hep10k_batch(graphs, seed=seed)
    raise ValueError(dataset)


def oracle_forward(model: str, batch, w, nthreads: int, numeric: str = "f32"):
    from oracle import oracle
    if numeric == "q6.10":
        return oracle.gin_forward_q(batch, [w], nthreads=nthreads)[0]
    fn = {"GIN": oracle.gin_forward, "GIN-VN": oracle.gin_forward, "GCN": oracle.gcn_forward, "GAT": oracle.gat_forward,
          "PNA": oracle.pna_forward, "DGN": oracle.dgn_forward}[model]
    return fn(batch, [w], nthreads=nthreads)


def effective_cpus() -> int:
    """CPUs this process can really use: the affinity mask, capped by the cgroup CPU quota (a container that sees 256
    hardware threads under a 16-CPU quota runs 256 OpenMP threads at half the speed of 32)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]  # cgroup v2
        if quota != "max":
            n = min(n, max(1, -(-int(quota) // int(period))))
    except (OSError, ValueError):
        try:
            quota = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())  # cgroup v1
            period = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if quota > 0 and period > 0:
                n = min(n, max(1, -(-quota // period)))
        except (OSError, ValueError):
            pass
    return n


def cpu_baseline(model, batch, w, budget_s: float = 15.0, numeric: str = "f32"):
    """The oracle (CPU restatement of the reference, kind='port') timed on this host's cores on a
    bounded sample of the same workload.  Returns (record, oracle logits of the sample)."""
    cores = effective_cpus()
    probe = batch.slice(0, min(128, batch.num_graphs))
    t0 = time.perf_counter()
    oracle_forward(model, probe, w, 1, numeric)
    t1 = time.perf_counter()
    rate1 = probe.num_graphs / (t1 - t0)
    # OpenMP over graphs scales far from linearly on a big host: measure the parallel rate on a small sample first,
    # then size the timed sample for about budget_s seconds of wall clock
    warm = batch.slice(0, min(batch.num_graphs, 16 * cores))
    t0 = time.perf_counter()
    oracle_forward(model, warm, w, cores, numeric)
    ratep = warm.num_graphs / (time.perf_counter() - t0)
    n = int(min(batch.num_graphs, max(256, ratep * budget_s)))
    sample = batch.slice(0, n)
    t0 = time.perf_counter()
    logits = oracle_forward(model, sample, w, cores, numeric)
    t1 = time.perf_counter()
    rec = {"value": n / (t1 - t0), "unit": "graphs/s", "cores": cores, "kind": "port",
           "sample": f"first {n} graphs of the bench batch, oracle/{'ginq' if numeric == 'q6.10' else model.lower().replace('-vn', '')}_oracle.c, "
                     f"OpenMP over graphs, {cores} threads",
           "single_core_value": rate1}
    return rec, np.asarray(logits, dtype=np.float32)


# parity of the timed batch against the oracle, the tolerance of the tests (tests/parity.py): one rule for every model,
# |gpu - oracle| <= 1e-4 x (scale + |oracle|), scale = max(1, the oracle's own largest magnitude) -- measured on the oracle's output, no literals
PARITY_REL = 1e-4


def parity_record(model, got, want, numeric="f32"):
    got, want = np.asarray(got, np.float32), np.asarray(want, np.float32)
    err = np.abs(got.astype(np.float64) - want.astype(np.float64))
    if numeric != "f32":
        return {"graphs": int(want.shape[0]), "max_abs_err": float(err.max()) if err.size else 0.0, "tol": "bit-exact (Q patterns)",
                "ok": bool(np.array_equal(got, want))}
    scale = max(1.0, float(np.abs(want).max())) if want.size else 1.0
    bound = PARITY_REL * (scale + np.abs(want))
    return {"graphs": int(want.shape[0]), "max_abs_err": float(err.max()) if err.size else 0.0,
            "max_abs_oracle": float(np.abs(want).max()) if want.size else 0.0,
            "tol": f"{PARITY_REL:g} x (max(1, max|oracle|) + |x|)",
            "ok": bool((err <= bound).all() and np.isfinite(got).all())}


def rooflines(model, M, prof, kern, G, N, E, steps, split, qmode, tiles=None):
    """(roofline of the dominant kernel, roofline of the stand-alone aggregation kernel) from the HIP-event profile of the timed
    region: algorithmic bytes / flops per launch over the kernel's average launch duration."""
    agg_bytes, mlp_flops = M["agg_bytes"](N, E), M["flops"](N, E)
    layer = {k: v for k, v in prof.items() if k in M["hbm_kernels"] + M["mfma_kernels"]}
    dominant = max(layer.items(), key=lambda kv: kv[1]["total_ms"])[0] if layer else None
    agg_name = next((k for k in M["hbm_kernels"] if k in kern), M["hbm_kernels"][0])
    traffic_db = {}
    try:
        tj = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))
        if tj.get(model, {}).get("graphs") == G:
            traffic_db = tj[model]
    except (OSError, ValueError):
        pass

    def traffic_of(name):  # PMC-measured HBM bytes per launch of the same batch (committed profile), or None
        return (traffic_db.get(name) or {}).get("bytes")

    def tag(obj):  # where `traffic` comes from: a committed rocprofv3 --pmc pass of the same batch, not this run
        if obj is not None and obj.get("traffic") is not None:
            obj["traffic_source"] = "profiles/traffic.json: committed rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of the same batch, not this run"
        return obj

    def hbm_obj(name):
        if name not in kern:
            return None
        nbytes = agg_bytes * M.get("layers_per_launch", {}).get(name, 1)
        ach = nbytes / (kern[name] * 1e-3) / 1e9
        obj = {"kernel": name, "bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
               "frac": ach / HBM_PEAK_GBS, "traffic": traffic_of(name), "avg_ms": kern[name], "bytes_per_launch": nbytes}
        if name in M.get("moved_bytes", {}):
            obj["hbm_bytes_moved"] = M["moved_bytes"][name](N, E, tiles or N // TILE_ROWS.get(model, 256) + 1)
        lim = issue_limits(model, name)
        if lim is not None:
            obj["issue_limit"] = lim
        return tag(obj)

    roof = None
    if dominant in M["hbm_kernels"]:
        roof = hbm_obj(dominant)
    elif dominant is not None:
        launches_per_step = prof[dominant]["launches"] / max(steps, 1)
        t_s = kern[dominant] * 1e-3
        # dense layers one launch really executes (GIN resident: the folded readout removes the last layer's second linear layer,
        # so 4.5 of the 5 layers' products are computed and counted)
        work_flops = mlp_flops * M.get("dense_layers_per_launch", M.get("layers_per_launch", {})).get(dominant, 1)
        if split:
            # three f16 products per algorithmic fp32 product, priced against the f16 pipe the kernel uses
            ach, peak = 3 * work_flops / t_s / 1e12, F16_MFMA_PEAK_TF
            mfma = {"bound": "mfma", "achieved": ach, "peak": peak, "unit": "TFLOP/s", "frac": ach / peak,
                    "pipe": "f16 (3 products per fp32 product, fp32 accumulate)", "flops_per_launch": 3 * work_flops,
                    "fp32_equivalent_tflops": work_flops / t_s / 1e12,
                    # the USEFUL share of the f16 pipe: algorithmic fp32 flops / f16 peak (the other two thirds of `frac` are the
                    # price of carrying fp32 accuracy through an f16 pipe)
                    "useful_frac": work_flops / t_s / 1e12 / peak}
        else:
            ach = work_flops / t_s / 1e12
            mfma = {"bound": "mfma", "achieved": ach, "peak": FP32_MFMA_PEAK_TF, "unit": "TFLOP/s",
                    "frac": ach / FP32_MFMA_PEAK_TF, "pipe": "f32", "flops_per_launch": work_flops}
        fbf = M.get("fused_bytes")
        fbf = fbf.get(dominant) if isinstance(fbf, dict) else fbf
        if split and fbf is not None and dominant not in M.get("mfma_bound_kernels", ()):
            fb = fbf(N, E)
            ach = fb / t_s / 1e9
            roof = {"kernel": dominant, "bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": ach / HBM_PEAK_GBS, "traffic": traffic_of(dominant), "avg_ms": kern[dominant],
                    "bytes_per_launch": fb, "mfma": mfma}
            if dominant in M.get("moved_bytes", {}):
                roof["hbm_bytes_moved"] = M["moved_bytes"][dominant](N, E, tiles or N // TILE_ROWS.get(model, 256) + 1)
        else:
            roof = dict({"kernel": dominant, "traffic": traffic_of(dominant), "avg_ms": kern[dominant],
                         "launches_per_step": launches_per_step}, **mfma)
            if fbf is not None:
                roof["hbm_bytes_per_launch"] = fbf(N, E)
            if dominant in M.get("moved_bytes", {}):
                roof["hbm_bytes_moved"] = M["moved_bytes"][dominant](N, E, tiles or N // TILE_ROWS.get(model, 256) + 1)
        tag(roof)
        lim = issue_limits(model, dominant)
        if lim is not None:
            roof["issue_limit"] = lim
    agg = hbm_obj(agg_name) if not qmode else None
    if qmode:
        roof = None  # integer VALU work (one truncated product at a time): neither of the two rooflines applies
    return roof, agg


def issue_limits(model, kernel):
    """What the kernel is really limited by, from the committed SQ counter pass of the same batch (profiles/limits.json, made by
    profiles/make_limits.py from profiles/rNN_<M>_pmc_SQ*.txt): busy share of the matrix pipe, the LDS and VALU issue."""
    try:
        lj = json.load(open(os.path.join(ROOT, "profiles", "limits.json")))
        rec = lj.get(model, {}).get(kernel)
        return dict(rec, source="profiles/limits.json (committed rocprofv3 --pmc SQ pass of the same batch, not this run)") if rec else None
    except (OSError, ValueError):
        return None


def tile_count(eng, model, batch):
    """Graph tiles the resident kernel walks for the engine's resident batch: the bin-packed count when the model packs (flowgnn_batch_tiles),
    else the batch-order count; None for a model / batch without graph tiles."""
    try:
        plain, packed = eng.batch_tiles()
    except Exception:
        return None
    return (packed or plain) or None


def measure_config(model, batch, steps, warmup, device, sample_graphs=512):
    """One of the non-headline configurations in the same process.  `value` / `ms_per_step` come from `steps` runs of the resident
    batch bracketed by stream syncs with NO profiling (what a caller gets: dataset-sized batches replay a hipGraph, which HIP events
    around every launch would switch off -- 167 vs 193 us per step at 4 113 graphs); the kernel times behind `frac` come from a
    second, HIP-event-profiled pass of the same number of steps (`ms_per_step_profiled`).  Then the stand-alone aggregation probe
    (`aggregation`), what the committed counter passes say limits the kernel (`limit`, `wait_share`, `hbm_bytes_moved`), and a parity
    check of the first `sample_graphs` graphs against the oracle.  Compact on purpose (one JSON line carries all of them)."""
    from flowgnn_amd import Engine, weights
    M = MODELS[model]
    w = weights.SYNTH[model](seed=7)
    eng = Engine(model, device=device)
    agg_ms = None
    try:
        eng.set_weights(w)
        eng.set_batch(batch)
        tiles = tile_count(eng, model, batch)
        for _ in range(warmup):
            eng.run()
        eng.sync()
        t0 = time.perf_counter()
        for _ in range(steps):
            eng.run()
        eng.sync()
        dt = time.perf_counter() - t0
        eng.profile_enable(True)
        t0 = time.perf_counter()
        for _ in range(steps):
            eng.run()
        eng.sync()
        dt_prof = time.perf_counter() - t0
        prof = eng.profile_read()
        eng.profile_enable(False)
        out = eng.results()
        reruns = eng.exact_reruns()
        try:
            agg_ms = eng.aggregation_only_ms(layer=0, iters=10)
        except Exception:  # a model without a stand-alone aggregation kernel (GAT: the layer IS the aggregation)
            agg_ms = None
    finally:
        eng.close()
    G, N, E = batch.num_graphs, batch.total_nodes, batch.total_edges
    kern = {k: (v["total_ms"] / max(v["launches"], 1)) for k, v in prof.items()}
    agg_name = next((k for k in M["hbm_kernels"] if k in kern), M["hbm_kernels"][0])
    if agg_ms is not None and agg_name not in kern:
        kern[agg_name] = agg_ms
    roof, agg = rooflines(model, M, prof, kern, G, N, E, steps, True, False, tiles)
    n = min(G, sample_graphs)
    want = np.asarray(oracle_forward(model, batch.slice(0, n), w, effective_cpus()), np.float32)
    par = parity_record(model, out[:n], want)
    rec = {"value": G * steps / dt, "ms_per_step": dt / steps * 1e3, "ms_per_step_profiled": dt_prof / steps * 1e3, "graphs": G,
           "kernel": roof["kernel"] if roof else None,
           "avg_ms": roof["avg_ms"] if roof else None, "bound": roof["bound"] if roof else None, "frac": roof["frac"] if roof else None,
           "parity_ok": par["ok"], "max_abs_err": par["max_abs_err"], "exact_reruns": reruns}
    if roof:
        lim = roof.get("issue_limit") or {}
        if lim:  # committed SQ counter pass of this kernel (profiles/limits.json): what binds, and how long its waves are parked
            rec["limit"] = {"what": lim.get("limit"), "busy": lim.get(lim.get("limit")), "wait_share": lim.get("wait_share"),
                            "mfma_busy": lim.get("mfma_busy"), "lds_busy": lim.get("lds_busy"), "valu_issue": lim.get("valu_issue"),
                            # matrix pipe + VALU of a SIMD share its issue (tools/coissue4.hip): their shares add up to <= ~1.2
                            "simd_issue": lim.get("simd_issue"), "simd_issue_ceiling": lim.get("simd_issue_ceiling")}
        moved = roof.get("traffic") if roof.get("traffic") is not None else roof.get("hbm_bytes_moved")
        if moved is not None:
            rec["hbm_bytes_moved"] = moved
            rec["hbm_bytes_moved_source"] = "pmc (profiles/traffic.json)" if roof.get("traffic") is not None else "formula"
        if roof.get("bound") == "hbm":
            rec["bytes_priced"] = roof.get("bytes_per_launch")
        if "useful_frac" in roof:
            rec["useful_frac"] = roof["useful_frac"]
    if agg and agg.get("kernel") != (roof or {}).get("kernel"):  # the message-passing unit alone (north_star's aggregation roofline)
        rec["aggregation"] = {"kernel": agg["kernel"], "avg_ms": agg["avg_ms"], "frac": agg["frac"], "bytes": agg["bytes_per_launch"]}
    return {k: (round(v, 4) if isinstance(v, float) and k not in ("value", "max_abs_err") else v) for k, v in rec.items()}


# ---------------------------------------------------------------------------------------------------------------------
# multi-GPU plumbing: one process per GPU, graphs are the only parallel dimension
# ---------------------------------------------------------------------------------------------------------------------
def plan_job(dataset: str, graphs: int, world: int, rank: int, scaling: str, make=None):
    """The graphs this rank carries in one step, and where they sit in the job.
    weak   : `graphs` graphs PER GPU, each rank its own synthetic shard (seed 1234 + rank): per-GPU work fixed as N grows.
    strong : ONE job batch of `graphs` graphs (seed 1234) cut into contiguous ranges balanced by sum(N + E)
             (flowgnn_amd.dist.shard_ranges); ragged shards, total work fixed.
    Returns (local batch, ranges [(g0, g1)] in job order, balance record or None)."""
    from flowgnn_amd import dist as fdist
    make = make or make_batch
    if scaling == "weak" or world == 1:
        return make(dataset, graphs, 1234 + rank), [(r * graphs, (r + 1) * graphs) for r in range(world)], None
    job = make(dataset, graphs, 1234)
    ranges = fdist.shard_ranges(job, world)
    work = job.nums_of_nodes.astype(np.int64) + job.nums_of_edges.astype(np.int64)
    loads = [int(work[a:b].sum()) for a, b in ranges]
    mean = sum(loads) / world
    g0, g1 = ranges[rank]
    return job.slice(g0, g1), ranges, {"graphs_per_rank": [b - a for a, b in ranges], "node_plus_edge_load_per_rank": loads,
                                       "imbalance_max_over_mean": max(loads) / mean if mean else 1.0,
                                       "job_nodes": int(job.total_nodes), "job_edges": int(job.total_edges),
                                       "_job_counts": (job.nums_of_nodes, job.nums_of_edges)}  # (popped before the record is printed)


class ShardedResults:
    """Result concat of the job: every rank's readout kernel writes its per-graph logits into `pad` (width = widest
    shard), ONE all_gather_into_tensor per step concatenates them (RCCL over xGMI on GPUs, gloo in the CPU test).
    Two buffer pairs alternate from step to step and the gather is asynchronous: step i + 1 computes into the other pair
    while step i's logits travel, and a pair is only reused once its gather has been waited for."""

    def __init__(self, ranges, rank, device, dist_mod, collectives=None):
        import torch
        self.ranges, self.rank, self.dist = ranges, rank, dist_mod
        self.world = len(ranges)
        # collectives: whether the per-step gather is issued at all (default: only with more than one rank; the one-GPU RCCL check
        # of tests/test_rccl_gpu.py forces it with a group of one)
        self.collectives = self.world > 1 if collectives is None else bool(collectives)
        self.width = max(1, max(b - a for a, b in ranges))
        nbuf = 2 if self.collectives else 1
        self.pads = [torch.zeros(self.width, dtype=torch.float32, device=device) for _ in range(nbuf)]
        self.alls = [torch.empty(self.world * self.width, dtype=torch.float32, device=device) for _ in range(nbuf)] if self.collectives else self.pads
        self.work = [None] * nbuf
        self.cur = 0    # the pair the coming step writes
        self.last = 0   # the pair the last finished step wrote

    @property
    def pad(self):
        """Where this rank's readout writes the coming step's logits (waits for the gather that last used the pair)."""
        if self.work[self.cur] is not None:
            self.work[self.cur].wait()
            self.work[self.cur] = None
        return self.pads[self.cur]

    def local_count(self):
        a, b = self.ranges[self.rank]
        return b - a

    def gather(self):
        self.last = self.cur
        if self.collectives:
            self.work[self.cur] = self.dist.all_gather_into_tensor(self.alls[self.cur], self.pads[self.cur], async_op=True)
            self.cur ^= 1

    def finish(self):
        for i, w in enumerate(self.work):
            if w is not None:
                w.wait()
                self.work[i] = None

    def local(self):
        return self.pads[self.last]

    def assemble(self):
        """Logits of the whole job in job order (ragged shards trimmed), from the last finished step."""
        import torch
        self.finish()
        rows = self.alls[self.last].view(self.world, self.width)
        return torch.cat([rows[r, : b - a] for r, (a, b) in enumerate(self.ranges)])


def launch_ranks(n: int, argv):
    """`python bench.py --gpus N` run plainly (no launcher): start the N ranks ourselves, one process per GPU."""
    import socket
    import subprocess
    import torch
    have = torch.cuda.device_count() if torch.cuda.is_available() else 0
    if have < n:
        raise SystemExit(f"bench.py: --gpus {n} requested but {have} GPU(s) visible; refusing to benchmark fewer GPUs than asked for")
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + list(argv), env=env))
    rc = 0
    live = list(procs)
    while live:
        for p in list(live):
            code = p.poll()
            if code is None:
                continue
            live.remove(p)
            if code != 0 and rc == 0:
                rc = code
                for q in live:  # one rank failed: stop exactly the processes we started
                    q.terminate()
        time.sleep(0.05)
    raise SystemExit(rc)


def dry_run(args):
    """`--dry-run`: the N-rank job of `--gpus N` planned and its result concat exercised in ONE process on CPU tensors -- plan_job for
    every rank, a ShardedResults per rank over a stand-in for the process group whose all-gather copies the ranks' pads (what RCCL
    does over xGMI) -- and the per-rank shard balance printed as one JSON line.  No GPU, no kernels: what can go wrong in planning
    (ragged ranges, pad width, job order of the concatenated logits) fails here, before the first multi-GPU run."""
    import torch
    M = MODELS[args.model]
    graphs, world = args.graphs or M["graphs"], args.gpus
    plans = [plan_job(M["dataset"], graphs, world, r, args.scaling) for r in range(world)]
    ranges = plans[0][1]
    assert all(p[1] == ranges for p in plans), "ranks disagree about the job's ranges"

    class Done:
        def wait(self):
            return True

    class OneProcessGroup:
        def __init__(self):
            self.members = []

        def all_gather_into_tensor(self, out, pad, async_op=True):
            out.copy_(torch.cat([m.pads[m.cur] for m in self.members]))
            return Done()

    group = OneProcessGroup()
    group.members = [ShardedResults(ranges, r, "cpu", group, collectives=True) for r in range(world)]
    for r, m in enumerate(group.members):  # every rank's readout "writes" the job-wide ids of its graphs
        a, b = ranges[r]
        assert plans[r][0].num_graphs == b - a == m.local_count(), (r, plans[r][0].num_graphs, a, b)
        m.pad[: b - a] = torch.arange(a, b, dtype=torch.float32)
    for m in list(group.members):
        # (gather() flips `cur`: every member gathers from the pads the others hold NOW, so flip them together afterwards)
        m.last = m.cur
        m.work[m.cur] = group.all_gather_into_tensor(m.alls[m.cur], m.pads[m.cur])
    total = ranges[-1][1]
    ok = all(torch.equal(m.assemble(), torch.arange(total, dtype=torch.float32)) for m in group.members)
    loads = [int(p[0].total_nodes + p[0].total_edges) for p in plans]
    mean = sum(loads) / world
    line = {"dry_run": True, "metric": M["metric"], "n_gpus": world, "scaling": args.scaling if world > 1 else "weak",
            "graphs_per_step_job": total, "pad_width": group.members[0].width,
            "ranks": [{"rank": r, "range": list(ranges[r]), "graphs": p[0].num_graphs, "nodes": p[0].total_nodes, "edges": p[0].total_edges,
                       "node_plus_edge_load": loads[r]} for r, p in enumerate(plans)],
            "imbalance_max_over_mean": max(loads) / mean if mean else 1.0, "result_concat_in_job_order": bool(ok)}
    print(json.dumps(line))
    if not ok:
        sys.exit(3)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--model", default="GIN", choices=sorted(MODELS))
    ap.add_argument("--graphs", type=int, default=0,
                    help="graphs per GPU per step (weak scaling; default: the model's roofline batch) or in the whole job (strong scaling)")
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"],
                    help="weak: fixed graphs per GPU; strong: ONE job batch cut by sum(N+E) with flowgnn_amd.dist.shard_ranges")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-entry-point", action="store_true", help="skip the timing of the drop-in symbol with host arrays (profiling runs: its launches would mix into the kernel trace)")
    ap.add_argument("--configs", default="auto", choices=["auto", "on", "off"],
                    help="also measure the other BASELINE configs (GIN at dataset size, GIN-VN, GCN, GAT, PNA, DGN) in the same run and "
                         "report them in a compact `configs` object; auto = on for the default single-GPU GIN run")
    ap.add_argument("--inject-parity-failure", action="store_true",
                    help="test hook: perturb the GPU logits before they are compared with the oracle (the run must then exit with code 3)")
    ap.add_argument("--numeric", default="f32", choices=["f32", "q6.10"],
                    help="q6.10: the reference's ap_fixed<16,6> bit-faithful mode (a fidelity mode, ~10x slower)")
    ap.add_argument("--dry-run", action="store_true",
                    help="plan the --gpus N job and exercise its result concat for all N ranks in this one process on CPU tensors; print the per-rank shard balance (no GPU needed)")
    args = ap.parse_args()
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if args.dry_run:
        return dry_run(args)

    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: what RCCL needs on this driver (already exported on the pool)
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        launch_ranks(args.gpus, sys.argv[1:])  # does not return
    # stdout carries ONE line, the JSON record.  Libraries write there too (RCCL prints "Librccl path : ..." on stdout when its
    # group comes up or goes down), so file descriptor 1 is pointed at stderr for the life of the process and the record is written
    # to a private duplicate of the original stdout.
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    import torch
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: refusing to report a number for a different GPU count")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the HIP path has no CPU fallback)")
    if torch.cuda.device_count() <= local_rank:
        raise SystemExit(f"rank {rank}: LOCAL_RANK={local_rank} but {torch.cuda.device_count()} GPU(s) visible")
    torch.cuda.set_device(local_rank)
    dist = None
    # FLOWGNN_BENCH_FORCE_COLLECTIVES=1: run the N > 1 code path (RCCL group, barriers, per-step all-gather, max over ranks) with a
    # group of ONE rank -- the only way to execute it on a one-GPU box (RCCL refuses two ranks on one device)
    collectives = world > 1 or os.environ.get("FLOWGNN_BENCH_FORCE_COLLECTIVES") == "1"
    if collectives:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))

    from flowgnn_amd import Engine, weights

    M = MODELS[args.model]
    graphs = args.graphs or M["graphs"]
    batch, ranges, balance = plan_job(M["dataset"], graphs, world, rank, args.scaling)
    w = weights.SYNTH[args.model](seed=7)
    eng = Engine(args.model, device=local_rank)
    eng.set_weights(w)
    if args.numeric != "f32":
        eng.set_numeric_mode(args.numeric)
    if balance:  # strong scaling: this rank holds a shard of ONE job -- kernels are chosen as one engine would for all of it
        eng.set_job_totals(balance["job_nodes"], balance["job_edges"])
        eng.set_job_tile_fill(eng.graph_tile_fill(*balance.pop("_job_counts")))  # ... and takes the job's side of the tile-fill threshold
    eng.set_batch(batch)
    G, N, E = batch.num_graphs, batch.total_nodes, batch.total_edges

    # The engine launches on the torch stream the collective orders itself against, so "forward, then all-gather" needs
    # no host synchronisation: the gather waits for the readout kernel on the device, and the next step's readout for the
    # gather (flowgnn_set_stream).
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):  # the pads are zero-filled on the stream that later writes them
        res = ShardedResults(ranges, rank, "cuda", dist, collectives)
    torch.cuda.synchronize()
    eng.set_stream(side.cuda_stream)
    bound = [0]

    def step():
        ptr = res.pad.data_ptr()  # the pair that is free: the previous step's logits may still be travelling
        if ptr != bound[0]:       # one GPU: set once; N GPUs: the two pairs alternate (the call itself never waits for the device)
            eng.set_results_buffer(ptr)
            bound[0] = ptr
        eng.run()
        res.gather()

    with torch.cuda.stream(side):
        for _ in range(args.warmup):
            step()
        eng.sync()
        eng.profile_enable(True)  # HIP events around every kernel launch, on the stream the kernels are launched on
        if collectives:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        eng.sync()  # inside the clock: stream sync + validation / range flags (an exact-fp32 re-run, if any, is timed too)
        res.finish()  # ... and the last gathers
        torch.cuda.synchronize()
        if collectives:
            dist.barrier()
        t1 = time.perf_counter()
        elapsed = t1 - t0
        if collectives:
            t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed = float(t.item())
        prof = eng.profile_read()
        eng.profile_enable(False)
        logits_all = res.assemble()
        ok = bool(torch.isfinite(logits_all).all().item())
        out_local = res.local()[:G].detach().cpu().numpy()
    total_job_graphs = ranges[-1][1]
    if int(logits_all.shape[0]) != total_job_graphs:
        raise SystemExit(f"result concat has {int(logits_all.shape[0])} graphs, the job has {total_job_graphs}")
    if collectives:  # what travelled is what this rank's readout wrote
        g0, g1 = ranges[rank]
        if not np.array_equal(logits_all[g0:g1].detach().cpu().numpy(), out_local):
            raise SystemExit(f"rank {rank}: its range of the gathered logits differs from what its readout wrote")

    exit_code = 0
    if rank == 0:
        value = total_job_graphs * args.steps / elapsed
        kern = {k: (v["total_ms"] / max(v["launches"], 1)) for k, v in prof.items()}
        agg_name = next((k for k in M["hbm_kernels"] if k in kern), M["hbm_kernels"][0])
        qmode = args.numeric != "f32"
        if agg_name not in kern and not qmode:  # fused layer: measure the message-passing unit alone as well
            try:
                kern[agg_name] = eng.aggregation_only_ms(layer=0, iters=10)
            except Exception:  # a model without a standalone aggregation kernel
                pass

        # the dense updates of every model run as three f16 MFMAs per fp32 product unless option <m>_mfma = 32 ("f32")
        split = eng.get_option({"GIN-VN": "gin"}.get(args.model, args.model.lower()) + "_mfma") != 32
        roof, agg = rooflines(args.model, M, prof, kern, G, N, E, args.steps, split, qmode, tile_count(eng, args.model, batch))
        par = f"batch-sharded x{world}, RCCL all-gather of logits" if world > 1 else "single GPU"
        if collectives and world == 1:
            par = "single GPU, the N > 1 code path forced (RCCL group of one rank: barriers, per-step all-gather, max over ranks)"
        line = {
            "metric": M["metric"],
            "value": value, "unit": "graphs/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": args.scaling if world > 1 else "weak",
            "vs_baseline": None, "dtype": "f32" if not qmode else "q6.10 (int16 patterns of ap_fixed<16,6>)", "data": "synthetic",
            "mfma_mode": ("f16x3-split (fp32-accurate; exact-f32 re-run on range overflow)" if split else "f32") if not qmode else None,
            "config": {"workload": M["workload"],
                       "graphs_per_step_job": total_job_graphs, "graphs_rank0": G, "nodes_rank0": N, "edges_rank0": E,
                       "parallelism": par},
            "finite": ok, "exact_reruns": eng.exact_reruns(),
            "roofline": roof,
            "aggregation_roofline": agg,
            "kernel_avg_ms": kern,
            "vs_fpga_u50": (value / FPGA_U50_GRAPHS_PER_S) if args.model == "GIN" else None,
        }
        if balance is not None:
            line["shard_balance"] = balance
        if world == 1 and not qmode and not args.no_entry_point:
            # the drop-in symbol itself with HOST arrays in the clock (validation, tile packing, PCIe copies, kernels, logits back):
            # reported beside `value`, never as `value`
            from flowgnn_amd import compute_graphs
            ts = []
            for _ in range(3):
                t0 = time.perf_counter()
                out_entry = compute_graphs(args.model, batch, [w])
                ts.append(time.perf_counter() - t0)
            line["entry_point_host_arrays"] = {"symbol": ("GIN" if args.model == "GIN-VN" else args.model) + "_compute_graphs", "ms": min(ts) * 1e3,
                                               "value": G / min(ts), "unit": "graphs/s",
                                               "matches_timed_logits": bool(np.allclose(out_entry, out_local, rtol=1e-5, atol=1e-5))}
        want = None
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"], want = cpu_baseline(args.model, batch, w, numeric=args.numeric)
            line["gpu_over_cpu"] = value / line["cpu_baseline"]["value"]
        else:  # no timed CPU leg: still check a slice of rank 0's shard of the timed batch against the oracle
            n = min(G, 4096)
            want = np.asarray(oracle_forward(args.model, batch.slice(0, n), w, effective_cpus(), args.numeric), np.float32)
        if args.inject_parity_failure:
            out_local = out_local + np.float32(1.0)
        line["parity"] = parity_record(args.model, out_local[: want.shape[0]], want, args.numeric)
        parity_ok = bool(line["parity"]["ok"])
        do_configs = args.configs == "on" or (args.configs == "auto" and world == 1 and args.model == "GIN" and not args.graphs and not qmode)
        if do_configs and world == 1:
            # every BASELINE config in the driver's one run: after the headline's timed region, same process, same GPU
            eng.close()
            eng = None
            from flowgnn_amd import graphpack as gp
            cfgs = {}
            # (warm-up long enough for the clocks to come back up after the seconds of host-side batch generation in between: with 2
            # warm-up steps the same kernels measured ~10 % slower than in a run of their own)
            csteps, cwarm = max(3, min(args.steps, 10)), 8
            mol = batch  # the headline's molhiv batch is reused for GAT and (plus virtual nodes) for GIN-VN
            # the dataset-sized batch: a step is 0.16 ms, so 50 steps behind 5 warm-up steps were 10 ms of GPU time in all -- shorter than
            # the clocks take to come back up after the host-side batch generation (0.180 ms per step measured that way, 0.155 in any
            # loop of 100 ms).  500 timed steps behind 300 of warm-up: 0.13 s.
            cfgs["GIN@4113"] = measure_config("GIN", make_batch("molhiv", 4113, 99), 500, 300, local_rank)
            cfgs["GAT"] = measure_config("GAT", mol, csteps, cwarm, local_rank)
            cfgs["GIN-VN"] = measure_config("GIN-VN", gp.add_virtual_nodes(mol), csteps, cwarm, local_rank)
            del mol
            cfgs["GCN"] = measure_config("GCN", make_batch("molpcba", MODELS["GCN"]["graphs"], 1234), csteps, cwarm, local_rank)
            hep = make_batch("hep10k", MODELS["PNA"]["graphs"], 1234)  # SURVEY 8(d)'s 2^16: every [N][D] fp32 tensor >= 1 GB
            cfgs["PNA"] = measure_config("PNA", hep, csteps, cwarm, local_rank)
            cfgs["DGN"] = measure_config("DGN", hep, csteps, cwarm, local_rank)
            half = hep.slice(0, 1 << 15)  # ... and half of it once, beside: how much of the stand-alone aggregation probe's rate is the
            del hep                       # 256 MiB Infinity Cache (the resident kernels keep h on chip: their rate should not move)
            cfgs["PNA@32768"] = measure_config("PNA", half, csteps, cwarm, local_rank)
            cfgs["DGN@32768"] = measure_config("DGN", half, csteps, cwarm, local_rank)
            line["configs"] = cfgs
            parity_ok = parity_ok and all(c["parity_ok"] for c in cfgs.values())
        os.write(json_fd, (json.dumps(line) + "\n").encode())
        if not parity_ok:
            sys.stderr.write("bench.py: PARITY FAILURE against the oracle -- the throughput above is not a valid measurement\n")
            exit_code = 3
    if eng is not None:
        eng.close()
    if collectives:  # (rank 0 leaves through the same barrier as the others even when it is about to report a failure)
        dist.barrier()
        dist.destroy_process_group()
    if exit_code:
        sys.exit(exit_code)


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""bench.py -- graphs/s of the GIN (dim 100) hot path on molhiv-shaped synthetic graphs.

    python bench.py --gpus N --steps K --warmup W [--graphs G_PER_GPU]

A "step" = one full batched forward of the resident batch through the C ABI (flowgnn_run):
batched load_graph (CSR build) -> atom encoder -> 5 x (aggregation + node MLP) -> mean-pool +
head, for G_PER_GPU graphs per GPU (weak scaling), followed for N > 1 by the RCCL all-gather that
concatenates the per-graph results.  Inputs are resident in HBM when the timed region starts (the
reference also times kernel execution only: run_experiments.sh:44).  Rank 0 prints ONE JSON line.

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


# Per-model bench table.  Algorithmic work per launch of the two kernel classes (DESIGN.md "Kernels"):
#   aggregation bytes (unpadded): read h once + write the aggregates + edge index (8 B) + edge payload
#   transform flops: 2 x MACs of the dense update per node per layer
MODELS = {
    "GIN": dict(metric="graphs/sec on ogbg-molhiv (GIN, dim=100)", dataset="molhiv", graphs=1 << 18,
                agg_bytes=lambda n, e: n * 400 * 2 + e * 20, flops=lambda n, e: n * 80000,
                # fused layer: read h once + write h' + CSR (row_ptr 4 B/node, src 4 B + edge code 1 B per edge)
                fused_bytes=lambda n, e: n * 400 * 2 + n * 4 + e * 5,
                hbm_kernels=("gin_aggregate",), mfma_kernels=("gin_layer_fused", "gin_mlp"),
                workload="GIN dim=100, batched ogbg-molhiv-shaped graphs on MI355X (BASELINE configs[1])"),
    "GIN-VN": dict(metric="graphs/sec on ogbg-molhiv (GIN-VN, dim=100)", dataset="molhiv-vn", graphs=1 << 18,
                   agg_bytes=lambda n, e: n * 400 * 2 + e * 20, flops=lambda n, e: n * 80000,
                   fused_bytes=lambda n, e: n * 400 * 2 + n * 4 + e * 5,
                   hbm_kernels=("gin_aggregate",), mfma_kernels=("gin_layer_fused", "gin_mlp"),
                   workload="GIN-VN dim=100 (virtual node per graph), ogbg-molhiv-shaped graphs"),
    "GCN": dict(metric="graphs/sec on ogbg-molpcba (GCN, dim=100)", dataset="molpcba", graphs=1 << 18,
                agg_bytes=lambda n, e: n * 400 * 2 + e * 24, flops=lambda n, e: n * 20000,
                # fused layer (aggregate + BN/root/degree epilogue + dense): rows in and out, CSR entry + norm + code per
                # edge, row bounds + out-degree per node; unfused dense layer: read a row, write a row
                fused_bytes={"gcn_layer_fused": lambda n, e: n * 400 * 2 + n * 8 + e * 9, "gcn_dense": lambda n, e: n * 400 * 2},
                hbm_kernels=("gcn_aggregate",), mfma_kernels=("gcn_layer_fused", "gcn_dense"),
                workload="GCN dim=100, batched ogbg-molpcba-shaped graphs on MI355X (BASELINE configs[2])"),
    "GAT": dict(metric="graphs/sec on ogbg-molhiv (GAT, 4 heads x 16)", dataset="molhiv", graphs=1 << 18,
                agg_bytes=lambda n, e: n * (256 + 32) + n * 256 + (e + n) * 8, flops=lambda n, e: n * 16384,
                hbm_kernels=("gat_layer",), mfma_kernels=(),
                workload="GAT 5-layer, 4 heads x 16, ogbg-molhiv-shaped graphs on MI355X (BASELINE configs[3])"),
    "PNA": dict(metric="graphs/sec on hep10k (PNA, dim=80)", dataset="hep10k", graphs=1 << 15,
                agg_bytes=lambda n, e: n * 320 + n * 320 * 4 + e * 8, flops=lambda n, e: n * 153600,
                fused_bytes=lambda n, e: n * (1280 + 320 + 320),  # split dense: read 4 aggregates + h, write h'
                hbm_kernels=("pna_aggregate",), mfma_kernels=("pna_dense",),
                workload="PNA dim=80, hep10k-shaped kNN graphs on MI355X (BASELINE configs[4])"),
    "DGN": dict(metric="graphs/sec on hep10k (DGN, dim=100)", dataset="hep10k", graphs=1 << 15,
                agg_bytes=lambda n, e: n * 400 * 3 + e * 12, flops=lambda n, e: n * 40000,
                fused_bytes=lambda n, e: n * (800 + 400 + 400),  # split dense: read both aggregates + h, write h'
                hbm_kernels=("dgn_aggregate",), mfma_kernels=("dgn_dense",),
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
    bounded sample of the same workload."""
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
    oracle_forward(model, sample, w, cores, numeric)
    t1 = time.perf_counter()
    return {"value": n / (t1 - t0), "unit": "graphs/s", "cores": cores, "kind": "port",
            "sample": f"first {n} graphs of the bench batch, oracle/{'ginq' if numeric == 'q6.10' else model.lower().replace('-vn', '')}_oracle.c, "
                      f"OpenMP over graphs, {cores} threads",
            "single_core_value": rate1}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--model", default="GIN", choices=sorted(MODELS))
    ap.add_argument("--graphs", type=int, default=0, help="graphs per GPU per step (default: the model's roofline batch)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--numeric", default="f32", choices=["f32", "q6.10"],
                    help="q6.10: the reference's ap_fixed<16,6> bit-faithful mode (GIN / GIN-VN only; a fidelity mode, ~10x slower)")
    args = ap.parse_args()

    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: what RCCL needs on this driver (already exported on the pool)
    import torch
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the HIP path has no CPU fallback)")
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))

    from flowgnn_amd import Engine, weights

    M = MODELS[args.model]
    graphs = args.graphs or M["graphs"]
    batch = make_batch(M["dataset"], graphs, seed=1234 + rank)  # each rank its own shard of the job
    w = weights.SYNTH[args.model](seed=7)
    eng = Engine(args.model, device=local_rank)
    eng.set_weights(w)
    if args.numeric != "f32":
        eng.set_numeric_mode(args.numeric)
    eng.set_batch(batch)
    G, N, E = batch.num_graphs, batch.total_nodes, batch.total_edges

    out_local = torch.empty(G, dtype=torch.float32, device="cuda")
    eng.set_results_buffer(out_local.data_ptr())
    out_all = torch.empty(G * world, dtype=torch.float32, device="cuda") if world > 1 else out_local

    def step():
        eng.run()
        if world > 1:
            eng.sync()  # results complete before the collective reads them
            dist.all_gather_into_tensor(out_all, out_local)
            torch.cuda.current_stream().synchronize()  # ... and the gather done before the next step's readout rewrites them

    for _ in range(args.warmup):
        step()
    eng.sync()
    eng.profile_enable(True)  # HIP events around every kernel launch, on the engine's stream
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    eng.sync()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    t1 = time.perf_counter()
    elapsed = t1 - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    prof = eng.profile_read()
    eng.profile_enable(False)

    ok = bool(torch.isfinite(out_all).all().item())

    if rank == 0:
        total_graphs = G * world * args.steps  # every rank carries the same number of graphs
        value = total_graphs / elapsed
        agg_bytes, mlp_flops = M["agg_bytes"](N, E), M["flops"](N, E)
        kern = {k: (v["total_ms"] / max(v["launches"], 1)) for k, v in prof.items()}
        layer = {k: v for k, v in prof.items() if k in M["hbm_kernels"] + M["mfma_kernels"]}
        dominant = max(layer.items(), key=lambda kv: kv[1]["total_ms"])[0] if layer else None
        agg_name = M["hbm_kernels"][0]
        qmode = args.numeric != "f32"
        if agg_name not in kern and not qmode:  # fused layer: measure the message-passing unit alone as well
            kern[agg_name] = eng.aggregation_only_ms(layer=0, iters=10)

        traffic_db = {}
        try:
            tj = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))
            if tj.get(args.model, {}).get("graphs") == G:
                traffic_db = tj[args.model]
        except (OSError, ValueError):
            pass

        def traffic_of(name):  # PMC-measured HBM bytes per launch of the same batch (committed profile), or None
            return (traffic_db.get(name) or {}).get("bytes")

        def hbm_obj(name):
            ach = agg_bytes / (kern[name] * 1e-3) / 1e9
            return {"kernel": name, "bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": ach / HBM_PEAK_GBS, "traffic": traffic_of(name), "avg_ms": kern[name], "bytes_per_launch": agg_bytes}

        # the dense updates of every model run as three f16 MFMAs per fp32 product unless FLOWGNN_<M>_MFMA=f32
        env = {"GIN": "FLOWGNN_GIN_MFMA", "GIN-VN": "FLOWGNN_GIN_MFMA", "GCN": "FLOWGNN_GCN_MFMA", "PNA": "FLOWGNN_PNA_MFMA",
               "DGN": "FLOWGNN_DGN_MFMA", "GAT": "FLOWGNN_GAT_MFMA"}.get(args.model)
        split = env is not None and os.environ.get(env, "") != "f32"
        roof = None
        if dominant in M["hbm_kernels"]:
            roof = hbm_obj(dominant)
        elif dominant is not None:
            t_s = kern[dominant] * 1e-3
            if split:
                # three f16 products per algorithmic fp32 product, priced against the f16 pipe the kernel uses
                ach, peak = 3 * mlp_flops / t_s / 1e12, F16_MFMA_PEAK_TF
                mfma = {"bound": "mfma", "achieved": ach, "peak": peak, "unit": "TFLOP/s", "frac": ach / peak,
                        "pipe": "f16 (3 products per fp32 product, fp32 accumulate)", "flops_per_launch": 3 * mlp_flops,
                        "fp32_equivalent_tflops": mlp_flops / t_s / 1e12}
            else:
                ach = mlp_flops / t_s / 1e12
                mfma = {"bound": "mfma", "achieved": ach, "peak": FP32_MFMA_PEAK_TF, "unit": "TFLOP/s",
                        "frac": ach / FP32_MFMA_PEAK_TF, "pipe": "f32", "flops_per_launch": mlp_flops}
            if split and "fused_bytes" in M:
                # with the f16 pipe the fused layer's HBM floor (0.7 ms) is above its MFMA floor (0.6 ms): HBM-bound
                fbf = M["fused_bytes"]
                fb = (fbf[dominant] if isinstance(fbf, dict) else fbf)(N, E)
                ach = fb / t_s / 1e9
                roof = {"kernel": dominant, "bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": ach / HBM_PEAK_GBS, "traffic": traffic_of(dominant), "avg_ms": kern[dominant],
                        "bytes_per_launch": fb, "mfma": mfma}
            else:
                roof = dict({"kernel": dominant, "traffic": None, "avg_ms": kern[dominant]}, **mfma)
        agg = hbm_obj(agg_name) if not qmode else None
        if qmode:
            roof = None  # integer VALU work (one truncated product at a time): neither of the two rooflines applies
        line = {
            "metric": M["metric"],
            "value": value, "unit": "graphs/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32" if not qmode else "q6.10 (int16 patterns of ap_fixed<16,6>)", "data": "synthetic",
            "mfma_mode": ("f16x3-split (fp32-accurate; exact-f32 re-run on range overflow)" if split else "f32") if not qmode else None,
            "config": {"workload": M["workload"],
                       "graphs_per_gpu_per_step": G, "nodes_per_gpu": N, "edges_per_gpu": E,
                       "parallelism": f"batch-sharded x{world}, RCCL all-gather of logits"},
            "finite": ok, "exact_reruns": eng.exact_reruns(),
            "roofline": roof,
            "aggregation_roofline": agg,
            "kernel_avg_ms": kern,
            "vs_fpga_u50": (value / FPGA_U50_GRAPHS_PER_S) if args.model == "GIN" else None,
        }
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(args.model, batch, w, numeric=args.numeric)
            line["gpu_over_cpu"] = value / line["cpu_baseline"]["value"]
        print(json.dumps(line), flush=True)
    eng.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

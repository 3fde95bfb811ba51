"""bench.py's own multi-GPU plumbing, driven on CPU: world-size-2 `gloo` run of plan_job (weak and strong scaling,
sum(N+E)-balanced ragged shards) + ShardedResults (padded all-gather, trim, job order), with the CPU oracle standing in
for the engine; the self-launcher's refusal to run on fewer GPUs than asked for; parity_record's verdicts."""
import os
import subprocess
import sys

import numpy as np
import torch.multiprocessing as mp

import bench

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, scaling, graphs, q):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    import bench as b
    from flowgnn_amd import weights
    from oracle import oracle
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    batch, ranges, balance = b.plan_job("hep10k-noeig" if scaling == "strong" else "molhiv", graphs, world, rank, scaling,
                                        make=_make)
    w = weights.synth_gin_weights(seed=7)
    res = b.ShardedResults(ranges, rank, "cpu", dist)
    assert res.local_count() == batch.num_graphs
    for _ in range(3):  # three "steps": both buffer pairs are used and one is reused (asynchronous gathers)
        if batch.num_graphs:  # (a rank with an empty shard computes nothing and still gathers)
            res.pad[: batch.num_graphs] = torch.from_numpy(oracle.gin_forward(batch, [w]))
        res.gather()
    q.put((rank, res.assemble().numpy(), ranges, balance))
    dist.barrier()
    dist.destroy_process_group()


def _make(dataset, graphs, seed):
    from flowgnn_amd import graphpack as gp
    if dataset == "hep10k-noeig":  # graph sizes vary a lot: the balance by sum(N + E) matters
        return gp.concat_batches([gp.synth_hep10k_batch(graphs // 2, seed=seed, with_eigen=False),
                                  gp.synth_molhiv_batch(graphs - graphs // 2, seed=seed)])
    return bench.make_batch(dataset, graphs, seed)


def _run(scaling, graphs, world=2):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31000 + (os.getpid() % 2000) + (7 if scaling == "strong" else 0) + 13 * world
    procs = [ctx.Process(target=_worker, args=(r, world, port, scaling, graphs, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = {}
    for _ in range(world):
        r, out, ranges, balance = q.get(timeout=300)
        res[r] = (out, ranges, balance)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    return res


def test_strong_scaling_shards_and_concat_world2(oracle, gin_weights):
    graphs = 41
    res = _run("strong", graphs)
    job = _make("hep10k-noeig", graphs, 1234)
    want = oracle.gin_forward(job, [gin_weights])
    for r in (0, 1):
        out, ranges, balance = res[r]
        assert np.array_equal(out, want)  # job order, ragged shards trimmed, identical on every rank
        assert ranges[0][0] == 0 and ranges[-1][1] == graphs and ranges[0][1] == ranges[1][0]
        assert ranges[0][1] - ranges[0][0] != ranges[1][1] - ranges[1][0]  # really ragged: cut by work, not by count
        assert balance["imbalance_max_over_mean"] < 1.2
        assert sum(balance["graphs_per_rank"]) == graphs


def test_strong_scaling_world8_with_fewer_graphs_than_ranks(oracle, gin_weights):
    """bench.py --gpus 8 --scaling strong on a job of SIX graphs: plan_job hands two ranks an empty shard and the others one graph
    each; ShardedResults pads to the widest shard, gathers, trims -- every rank assembles the job in job order."""
    graphs, world = 6, 8
    res = _run("strong", graphs, world)
    job = _make("hep10k-noeig", graphs, 1234)
    want = oracle.gin_forward(job, [gin_weights])
    sizes = None
    for r in range(world):
        out, ranges, balance = res[r]
        assert np.array_equal(out, want)
        assert len(ranges) == world and ranges[0][0] == 0 and ranges[-1][1] == graphs
        sizes = [c - a for a, c in ranges]
        assert sum(balance["graphs_per_rank"]) == graphs
    assert 0 in sizes and 1 in sizes


def test_weak_scaling_each_rank_its_own_shard_world2(oracle, gin_weights):
    res = _run("weak", 9)
    want = np.concatenate([oracle.gin_forward(bench.make_batch("molhiv", 9, 1234 + r), [gin_weights]) for r in range(2)])
    for r in (0, 1):
        out, ranges, balance = res[r]
        assert np.array_equal(out, want)
        assert ranges == [(0, 9), (9, 18)] and balance is None


def test_bench_refuses_fewer_gpus_than_asked_for():
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "64", "--steps", "1", "--warmup", "0"],
                       env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode != 0
    assert "--gpus 64" in p.stderr and "visible" in p.stderr
    assert "graphs/s" not in p.stdout  # no number for a GPU count that was not there
    # a launcher that started a different number of ranks than --gpus is refused as well
    env2 = dict(env, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4"], env=env2, capture_output=True, text=True, timeout=300)
    assert p.returncode != 0 and "WORLD_SIZE=2" in p.stderr


def test_parity_record():
    want = np.array([1.0, -2.0, 3.0], np.float32)
    ok = bench.parity_record("GIN", want + np.float32(5e-5), want)
    assert ok["ok"] and ok["graphs"] == 3 and 4e-5 < ok["max_abs_err"] < 6e-5
    assert not bench.parity_record("GIN", want + np.float32(1e-3), want)["ok"]
    assert not bench.parity_record("GIN", np.array([1.0, np.nan, 3.0], np.float32), want)["ok"]
    q = bench.parity_record("GIN", want, want, numeric="q6.10")
    assert q["ok"] and not bench.parity_record("GIN", want + np.float32(2 ** -10), want, numeric="q6.10")["ok"]


def test_dry_run_plans_every_rank_in_one_process():
    """`bench.py --gpus N --dry-run` (no GPU): the N-rank job planned and its result concat exercised in one process -- weak and strong
    scaling at 1 / 2 / 8 ranks, ragged strong-scaling shards included -- prints the per-rank shard balance."""
    import json
    for args in (["--gpus", "8", "--scaling", "strong", "--graphs", "3001"], ["--gpus", "2", "--graphs", "500", "--model", "DGN"],
                 ["--gpus", "1", "--graphs", "64"], ["--gpus", "8", "--scaling", "strong", "--graphs", "5"]):
        p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--dry-run"] + args, capture_output=True, text=True, timeout=300)
        assert p.returncode == 0, p.stderr[-2000:]
        d = json.loads(p.stdout.strip().splitlines()[-1])
        n = int(args[1])
        assert d["dry_run"] and d["n_gpus"] == n and d["result_concat_in_job_order"] and len(d["ranks"]) == n
        total = int(args[args.index("--graphs") + 1]) * (n if "strong" not in args else 1)
        assert d["graphs_per_step_job"] == total == sum(r["graphs"] for r in d["ranks"])
        assert [r["range"][0] for r in d["ranks"]] == [0] + [r["range"][1] for r in d["ranks"]][:-1]
        if "strong" in args and total >= 1000:
            assert d["imbalance_max_over_mean"] < 1.05

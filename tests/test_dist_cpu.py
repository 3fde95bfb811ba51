"""World-size-2 and world-size-8 gloo tests of the batch sharding + result concat (no GPU): the per-shard compute is
the CPU oracle here; on the GPU box the same function is fed Engine.forward.  At eight ranks the job has FEWER graphs than ranks:
empty and one-graph shards take part in the collective like any other."""
import os
import sys

import numpy as np
import torch.multiprocessing as mp

from flowgnn_amd import graphpack as gp, weights
from flowgnn_amd.dist import shard_ranges

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shard_ranges_cover_and_balance():
    b = gp.synth_hep10k_batch(40, seed=3, with_eigen=False)
    for ws in (1, 2, 3, 8, 64):
        r = shard_ranges(b, ws)
        assert r[0][0] == 0 and r[-1][1] == b.num_graphs
        assert all(r[i][1] == r[i + 1][0] for i in range(ws - 1))
    work = (b.nums_of_nodes + b.nums_of_edges).astype(np.int64)
    loads = [int(work[a:c].sum()) for a, c in shard_ranges(b, 4)]
    assert max(loads) - min(loads) <= 2 * int(work.max())


def _worker(rank, world, port, q, graphs=37):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from flowgnn_amd.dist import sharded_forward
    from oracle import oracle
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    b = gp.synth_molhiv_batch(graphs, seed=17)  # ragged split
    w = weights.synth_gin_weights(seed=7)
    out = sharded_forward(lambda shard: oracle.gin_forward(shard, [w]), b, rank, world)
    q.put((rank, out))
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_forward_gloo_world2(oracle, gin_weights):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    b = gp.synth_molhiv_batch(37, seed=17)
    want = oracle.gin_forward(b, [gin_weights])
    assert np.array_equal(res[0], want) and np.array_equal(res[1], want)


def test_sharded_forward_gloo_world8_with_empty_and_one_graph_shards(oracle, gin_weights):
    graphs, world = 5, 8
    b = gp.synth_molhiv_batch(graphs, seed=17)
    sizes = [c - a for a, c in shard_ranges(b, world)]
    assert sum(sizes) == graphs and 0 in sizes and 1 in sizes
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + 2000 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, world, port, q, graphs)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=300) for _ in range(world))
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    want = oracle.gin_forward(b, [gin_weights])
    assert all(np.array_equal(res[r], want) for r in range(world))

"""Batch-of-graphs data parallelism: the only parallelism the path has.

Graphs are independent (the reference iterates them one by one, GIN/src/GIN_compute.cc:44), so a
batch is cut into contiguous graph ranges balanced by sum(N + E); every rank runs the full model on
its range and the per-graph results are concatenated with ONE all-gather (RCCL over xGMI on GPUs,
gloo in the CPU tests).  There is no exchange inside the layers.
"""
from __future__ import annotations

from typing import Callable, List, Tuple

import numpy as np

from .graphpack import GraphBatch


def shard_ranges(batch: GraphBatch, world_size: int) -> List[Tuple[int, int]]:
    """Contiguous [g0, g1) per rank, balanced by cumulative node+edge count."""
    G = batch.num_graphs
    work = batch.nums_of_nodes.astype(np.int64) + batch.nums_of_edges.astype(np.int64)
    cum = np.concatenate([[0], np.cumsum(work)])
    total = int(cum[-1])
    cuts = [0]
    for r in range(1, world_size):
        # first g with cum[g] * world_size >= total * r: exact integers, the same rule as the C ABI's flowgnn_shard_ranges
        g = int(np.searchsorted(cum * world_size, total * r, side="left"))
        cuts.append(min(max(g, cuts[-1]), G))
    cuts.append(G)
    return [(cuts[r], cuts[r + 1]) for r in range(world_size)]


def sharded_forward(compute: Callable[[GraphBatch], np.ndarray], batch: GraphBatch, rank: int, world_size: int,
                    device: str = "cpu", group=None) -> np.ndarray:
    """Run `compute` on this rank's shard and all-gather the per-graph results (ragged shards are
    padded to the largest shard for the collective, then trimmed)."""
    import torch
    import torch.distributed as dist

    ranges = shard_ranges(batch, world_size)
    g0, g1 = ranges[rank]
    # (an empty range -- more ranks than graphs -- computes nothing and still takes part in the collective)
    local = np.asarray(compute(batch.slice(g0, g1)), dtype=np.float32) if g1 > g0 else np.zeros(0, np.float32)
    if world_size == 1:
        return local
    width = max(b - a for a, b in ranges)
    buf = torch.zeros(width, dtype=torch.float32, device=device)
    buf[: local.shape[0]] = torch.from_numpy(local).to(device)
    gathered = torch.empty(world_size * width, dtype=torch.float32, device=device)
    dist.all_gather_into_tensor(gathered, buf, group=group)
    g = gathered.cpu().numpy().reshape(world_size, width)
    return np.concatenate([g[r, : b - a] for r, (a, b) in enumerate(ranges)])

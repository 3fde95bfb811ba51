"""Graph packs in FlowGNN's on-disk format, plus seeded synthetic generators.

The reference ships no graphs (graphs/*.zip are missing blobs); the format is defined
only by its readers:

  graphs/graph_info/g%d_info.txt          "%d\\n%d" = num_of_nodes, num_of_edges   GIN/src/host.cc:14-15,129-131
  graphs/graph_bin/g%d_node_feature.bin   int32[N][9]                               GIN/src/host_load.cc:119-125
  graphs/graph_bin/g%d_edge_list.bin      int32[E][2]  (u, v), ids local to graph   GIN/src/host_load.cc:127-133
  graphs/graph_bin/g%d_edge_attr.bin      int32[E][3]                               GIN/src/host_load.cc:135-141
  DGN/eig/g%d.txt                         text dump of an N x 4 tensor              DGN/src/host_load.cc:178,201-215

Graph numbering is 1-based.  A *batch* is the concatenation the reference host builds
(GIN/src/host.cc:119-138): flat node_feature / edge_list / edge_attr arrays with LOCAL node
ids, plus nums_of_nodes / nums_of_edges.
"""
from __future__ import annotations

import os
from dataclasses import dataclass
from typing import Optional

import numpy as np

ND_FEATURE_CARD = np.array([119, 4, 12, 12, 10, 6, 6, 2, 2], dtype=np.int64)  # GIN/src/host_load.cc:5
ED_FEATURE_CARD = np.array([5, 6, 2], dtype=np.int64)  # GIN/src/host_load.cc:6


@dataclass
class GraphBatch:
    nums_of_nodes: np.ndarray  # int32 [G]
    nums_of_edges: np.ndarray  # int32 [G]
    node_feature: np.ndarray  # int32 [N_tot, 9]
    edge_list: np.ndarray  # int32 [E_tot, 2] local ids
    edge_attr: np.ndarray  # int32 [E_tot, 3]
    node_eigen: Optional[np.ndarray] = None  # float32 [N_tot, 4] (DGN)

    @property
    def num_graphs(self) -> int:
        return int(self.nums_of_nodes.shape[0])

    @property
    def total_nodes(self) -> int:
        return int(self.nums_of_nodes.sum())

    @property
    def total_edges(self) -> int:
        return int(self.nums_of_edges.sum())

    def node_offsets(self) -> np.ndarray:
        off = np.zeros(self.num_graphs + 1, dtype=np.int64)
        np.cumsum(self.nums_of_nodes, out=off[1:])
        return off

    def edge_offsets(self) -> np.ndarray:
        off = np.zeros(self.num_graphs + 1, dtype=np.int64)
        np.cumsum(self.nums_of_edges, out=off[1:])
        return off

    def slice(self, g0: int, g1: int) -> "GraphBatch":
        """Contiguous graph range [g0, g1) as its own batch (used for GPU sharding)."""
        no, eo = self.node_offsets(), self.edge_offsets()
        return GraphBatch(
            self.nums_of_nodes[g0:g1].copy(),
            self.nums_of_edges[g0:g1].copy(),
            self.node_feature[no[g0]:no[g1]].copy(),
            self.edge_list[eo[g0]:eo[g1]].copy(),
            self.edge_attr[eo[g0]:eo[g1]].copy(),
            None if self.node_eigen is None else self.node_eigen[no[g0]:no[g1]].copy(),
        )

    def global_edges(self) -> np.ndarray:
        """Edge list with global (batch-wide) node ids, int64 [E_tot, 2]."""
        no = self.node_offsets()
        gid = np.repeat(np.arange(self.num_graphs), self.nums_of_edges)
        return self.edge_list.astype(np.int64) + no[gid][:, None]


def concat_batches(batches) -> GraphBatch:
    eig = None
    if all(b.node_eigen is not None for b in batches):
        eig = np.concatenate([b.node_eigen for b in batches])
    return GraphBatch(
        np.concatenate([b.nums_of_nodes for b in batches]),
        np.concatenate([b.nums_of_edges for b in batches]),
        np.concatenate([b.node_feature for b in batches]),
        np.concatenate([b.edge_list for b in batches]),
        np.concatenate([b.edge_attr for b in batches]),
        eig,
    )


# --------------------------------------------------------------------------- on-disk format
def write_pack(batch: GraphBatch, root: str, eig_dir: Optional[str] = None) -> None:
    """Write `batch` as root/graph_info/g%d_info.txt + root/graph_bin/g%d_*.bin (1-based)."""
    os.makedirs(os.path.join(root, "graph_info"), exist_ok=True)
    os.makedirs(os.path.join(root, "graph_bin"), exist_ok=True)
    no, eo = batch.node_offsets(), batch.edge_offsets()
    for g in range(batch.num_graphs):
        n, e = int(batch.nums_of_nodes[g]), int(batch.nums_of_edges[g])
        with open(os.path.join(root, "graph_info", f"g{g + 1}_info.txt"), "w") as f:
            f.write(f"{n}\n{e}")
        base = os.path.join(root, "graph_bin", f"g{g + 1}")
        batch.node_feature[no[g]:no[g + 1]].astype("<i4").tofile(base + "_node_feature.bin")
        batch.edge_list[eo[g]:eo[g + 1]].astype("<i4").tofile(base + "_edge_list.bin")
        batch.edge_attr[eo[g]:eo[g + 1]].astype("<i4").tofile(base + "_edge_attr.bin")
        if eig_dir is not None and batch.node_eigen is not None:
            os.makedirs(eig_dir, exist_ok=True)
            rows = batch.node_eigen[no[g]:no[g + 1]]
            with open(os.path.join(eig_dir, f"g{g + 1}.txt"), "w") as f:
                f.write("tensor([" + ",\n        ".join(
                    "[" + ", ".join(f"{x: .4e}" for x in r) + "]" for r in rows) + "])\n")
    with open(os.path.join(root, "dataset_size.txt"), "w") as f:
        f.write(str(batch.num_graphs))


def read_pack(root: str, num_graphs: Optional[int] = None, eig_dir: Optional[str] = None) -> GraphBatch:
    """Read a pack written in the reference layout.  `num_graphs` defaults to
    root/dataset_size.txt (the reference compiles it in: common/includes/dataset/dataset.hpp)."""
    if num_graphs is None:
        with open(os.path.join(root, "dataset_size.txt")) as f:
            num_graphs = int(f.read().strip())
    nn, ne, nf, el, ea, eig = [], [], [], [], [], []
    for g in range(1, num_graphs + 1):
        with open(os.path.join(root, "graph_info", f"g{g}_info.txt")) as f:
            n, e = (int(x) for x in f.read().split()[:2])
        base = os.path.join(root, "graph_bin", f"g{g}")
        nf.append(np.fromfile(base + "_node_feature.bin", dtype="<i4", count=n * 9).reshape(n, 9))
        el.append(np.fromfile(base + "_edge_list.bin", dtype="<i4", count=e * 2).reshape(e, 2))
        p = base + "_edge_attr.bin"
        ea.append(np.fromfile(p, dtype="<i4", count=e * 3).reshape(e, 3) if os.path.exists(p)
                  else np.zeros((e, 3), dtype=np.int32))
        nn.append(n)
        ne.append(e)
        if eig_dir is not None:
            eig.append(_read_eig_txt(os.path.join(eig_dir, f"g{g}.txt"), n))
    return GraphBatch(
        np.asarray(nn, dtype=np.int32), np.asarray(ne, dtype=np.int32),
        np.concatenate(nf).astype(np.int32) if nf else np.zeros((0, 9), np.int32),
        np.concatenate(el).astype(np.int32) if el else np.zeros((0, 2), np.int32),
        np.concatenate(ea).astype(np.int32) if ea else np.zeros((0, 3), np.int32),
        np.concatenate(eig).astype(np.float32) if eig else None,
    )


def _read_eig_txt(path: str, n: int) -> np.ndarray:
    """DGN/src/host_load.cc:201-215 scans the text for numbers, 4 per node."""
    import re
    with open(path) as f:
        txt = f.read()
    vals = re.findall(r"[-+]?(?:\d+\.\d*|\.\d+|\d+)(?:[eE][-+]?\d+)?", txt)
    arr = np.asarray([float(v) for v in vals[: n * 4]], dtype=np.float32)
    return arr.reshape(n, 4)


# --------------------------------------------------------------------------- synthetic generators
def _features(rng: np.random.Generator, n_tot: int, e_und: int):
    nf = (rng.random((n_tot, 9)) * ND_FEATURE_CARD).astype(np.int32)
    ea = (rng.random((e_und, 3)) * ED_FEATURE_CARD).astype(np.int32)
    return nf, ea


def synth_molecule_batch(num_graphs: int, seed: int = 1234, mean_nodes: float = 25.27,
                         edges_per_node: float = 2.20, min_nodes: int = 6, max_nodes: int = 183) -> GraphBatch:
    """molhiv / molpcba-shaped graphs (GIN/src/dcl.h:39-45 gives 6/25/183 nodes).

    Each graph is a random chain-like tree (parent within the 3 previous atoms) plus ring-closure
    bonds (to an atom 4-5 positions back) so that directed E/N ~= edges_per_node; every bond is
    stored as (a, b) then (b, a) with the same attributes, as OGB's smiles2graph does.
    Fully vectorised: 2^18 graphs take a few seconds.
    """
    rng = np.random.default_rng(seed)
    sigma = 0.42
    mu = np.log(mean_nodes) - 0.5 * sigma * sigma + 0.012  # small correction for the clipping
    n = np.clip(np.rint(rng.lognormal(mu, sigma, num_graphs)), min_nodes, max_nodes).astype(np.int64)
    n_tot = int(n.sum())
    noff = np.zeros(num_graphs + 1, dtype=np.int64)
    np.cumsum(n, out=noff[1:])
    gid_node = np.repeat(np.arange(num_graphs), n)
    local = np.arange(n_tot) - noff[gid_node]

    # tree bonds: node i >= 1 -> parent in [i-3, i-1]
    child = np.nonzero(local >= 1)[0]
    back = 1 + (rng.random(child.size) * np.minimum(local[child], 3)).astype(np.int64)
    t_a = local[child] - back
    t_b = local[child]
    t_g = gid_node[child]

    # ring closures: r_g per graph so that total undirected bonds ~= edges_per_node/2 * n
    want = np.rint(edges_per_node * 0.5 * n).astype(np.int64)
    r = np.maximum(want - (n - 1), 0)
    r = np.where(n >= 6, r, 0)
    r_g = np.repeat(np.arange(num_graphs), r)
    hi = 5 + (rng.random(r_g.size) * (n[r_g] - 5)).astype(np.int64)  # in [5, n-1]
    span = 4 + (rng.random(r_g.size) * 2).astype(np.int64)  # 4 or 5 back: 5/6-rings
    r_a = hi - span
    r_b = hi

    und_g = np.concatenate([t_g, r_g])
    und_a = np.concatenate([t_a, r_a])
    und_b = np.concatenate([t_b, r_b])
    # keep graphs contiguous; inside a graph: tree bonds in atom order, then ring closures
    order = np.argsort(und_g, kind="stable")
    und_g, und_a, und_b = und_g[order], und_a[order], und_b[order]
    e_und = und_g.size
    nf, ea_und = _features(rng, n_tot, e_und)

    edge_list = np.empty((2 * e_und, 2), dtype=np.int32)
    edge_list[0::2, 0] = und_a
    edge_list[0::2, 1] = und_b
    edge_list[1::2, 0] = und_b
    edge_list[1::2, 1] = und_a
    edge_attr = np.repeat(ea_und, 2, axis=0)
    ne = 2 * np.bincount(und_g, minlength=num_graphs)
    return GraphBatch(n.astype(np.int32), ne.astype(np.int32), nf, edge_list, edge_attr.astype(np.int32))


def synth_molhiv_batch(num_graphs: int, seed: int = 1234) -> GraphBatch:
    """ogbg-molhiv shape: mean 25.27 nodes, 55.59 directed edges (GIN/summary.molhiv.csv:87-93)."""
    return synth_molecule_batch(num_graphs, seed, mean_nodes=25.27, edges_per_node=2.20)


def synth_molpcba_batch(num_graphs: int, seed: int = 1234) -> GraphBatch:
    """ogbg-molpcba shape: mean 26.99 nodes, 59.32 directed edges (GIN/summary.molpcba.csv)."""
    return synth_molecule_batch(num_graphs, seed, mean_nodes=26.99, edges_per_node=2.198, max_nodes=332)


def synth_hep10k_batch(num_graphs: int, seed: int = 1234, k: int = 16, with_eigen: bool = True) -> GraphBatch:
    """hep10k shape: N ~ 49 (20..100), k = 16 nearest neighbours on random 2-D points, so
    E = 16 N directed edges (GIN/summary.hep10k.csv: 491 322 nodes / 7 852 584 edges / 10 000 graphs).
    Edge (u, v): u is one of v's k nearest neighbours (message flows neighbour -> centre)."""
    rng = np.random.default_rng(seed)
    n = np.clip(np.rint(rng.normal(49.13, 12.0, num_graphs)), 20, 100).astype(np.int64)
    n_tot = int(n.sum())
    els = []
    for size in np.unique(n):
        idx = np.nonzero(n == size)[0]
        pts = rng.random((idx.size, size, 2))
        d = ((pts[:, :, None, :] - pts[:, None, :, :]) ** 2).sum(-1)
        d[:, np.arange(size), np.arange(size)] = np.inf
        kk = min(k, size - 1)
        nbr = np.argsort(d, axis=2, kind="stable")[:, :, :kk]  # [B, size(v), kk(u)]
        v = np.broadcast_to(np.arange(size)[None, :, None], nbr.shape)
        el = np.stack([nbr, v], axis=-1).reshape(idx.size, size * kk, 2)
        for j, g in enumerate(idx):
            els.append((g, el[j]))
    els.sort(key=lambda t: t[0])
    edge_list = np.concatenate([e for _, e in els]).astype(np.int32)
    ne = np.asarray([e.shape[0] for _, e in els], dtype=np.int32)
    nf, ea = _features(rng, n_tot, edge_list.shape[0])
    eig = None
    if with_eigen:
        eig = rng.uniform(-1.0, 1.0, (n_tot, 4)).astype(np.float32)
        noff = np.zeros(num_graphs + 1, dtype=np.int64)
        np.cumsum(n, out=noff[1:])
        gid = np.repeat(np.arange(num_graphs), n)
        norm = np.sqrt(np.add.reduceat(eig.astype(np.float64) ** 2, noff[:-1], axis=0))[gid]
        eig = (eig / np.maximum(norm, 1e-12)).astype(np.float32)
    return GraphBatch(n.astype(np.int32), ne, nf, edge_list, ea.astype(np.int32), eig)


def add_virtual_nodes(batch: GraphBatch) -> GraphBatch:
    """GIN-VN host augmentation (GIN-VN/src/host_load.cc:125-153, host.cc:133-134): per graph
    append node N with all-zero features, and after the real edges append (nd, N) and (N, nd) for
    every real node nd, attributes {0, 0, 0}."""
    no, eo = batch.node_offsets(), batch.edge_offsets()
    nfs, els, eas = [], [], []
    for g in range(batch.num_graphs):
        n = int(batch.nums_of_nodes[g])
        nfs.append(batch.node_feature[no[g]:no[g + 1]])
        nfs.append(np.zeros((1, 9), dtype=np.int32))
        els.append(batch.edge_list[eo[g]:eo[g + 1]])
        nd = np.arange(n, dtype=np.int32)
        extra = np.empty((2 * n, 2), dtype=np.int32)
        extra[0::2, 0] = nd
        extra[0::2, 1] = n
        extra[1::2, 0] = n
        extra[1::2, 1] = nd
        els.append(extra)
        eas.append(batch.edge_attr[eo[g]:eo[g + 1]])
        eas.append(np.zeros((2 * n, 3), dtype=np.int32))
    return GraphBatch(
        (batch.nums_of_nodes + 1).astype(np.int32),
        (batch.nums_of_edges + 2 * batch.nums_of_nodes).astype(np.int32),
        np.concatenate(nfs), np.concatenate(els), np.concatenate(eas), None)

"""Producer side of the reference's file formats -- the part the reference repo does not ship.

The reference consumes raw float32 `.bin` weight files (GIN/src/host_load.cc:24-58, GCN/src/host_load.cc:31-170) and a
per-graph pack (`graph_info/g%d_info.txt`, `graph_bin/g%d_*.bin`, GIN/src/host_load.cc:100-143).  Its GCN file is, float for
float, the flattened `state_dict` of the OGB `GNN(gnn_type='gcn', num_layer=5, emb_dim=100)` example model (BatchNorm blocks
of 4 x 100 floats + the `num_batches_tracked` counter: the 401-float stride of GCN/src/host_load.cc:118-166); its GIN files
are the same model family trained without BatchNorm.  This module writes those files from

  * a PyTorch `state_dict` (tensors or arrays) with the OGB example's parameter names -- for GIN, BatchNorm layers (inference
    statistics) are folded into the adjacent linear layers, which is exact, because the reference's GIN has no BatchNorm;
  * any sequence of graph objects with PyG `Data` attributes (`x`, `edge_index`, `edge_attr`, optional `eig`).

Nothing here needs torch, torch_geometric or ogb to be installed: tensors are taken through `numpy()` when they have it.
Host-side tooling only; the device path never imports this module.
"""
from __future__ import annotations

from collections import OrderedDict
from typing import Dict, Iterable, Mapping, Optional

import numpy as np

from . import weights as _weights
from .graphpack import GraphBatch, write_pack

ATOM_DIMS = (119, 4, 12, 12, 10, 6, 6, 2, 2)  # OGB AtomEncoder tables = offsets {0,119,123,...,171} of GIN/src/host_load.cc:5
BOND_DIMS = (5, 6, 2)                          # OGB BondEncoder tables = offsets {0,5,11} of GIN/src/message_passing.cc:3
BN_EPS = 1e-5                                  # torch.nn.BatchNorm1d default
REF_BN_EPS = 2.0 ** -10                        # what the reference adds to var (GCN/src/load_inputs.cc:32)


class ExportError(ValueError):
    pass


def _np(t) -> np.ndarray:
    if hasattr(t, "detach"):
        t = t.detach().cpu().numpy()
    return np.asarray(t, dtype=np.float64)


def _get(sd: Mapping, key: str) -> np.ndarray:
    if key not in sd:
        raise ExportError(f"state_dict has no '{key}'")
    return _np(sd[key])


def _tables(sd: Mapping, prefix: str, dims) -> np.ndarray:
    """Concatenate the per-feature embedding tables of an OGB Atom/BondEncoder into the reference's single table."""
    rows = []
    for k, n in enumerate(dims):
        t = _get(sd, f"{prefix}.{k}.weight")
        if t.shape[0] != n:
            raise ExportError(f"{prefix}.{k}.weight has {t.shape[0]} rows, the reference's offsets need {n}")
        rows.append(t)
    return np.concatenate(rows, axis=0)


def _fold_bn(w: np.ndarray, b: np.ndarray, sd: Mapping, prefix: str):
    """BatchNorm(inference)(W x + b) as one linear layer (W', b')."""
    g, beta = _get(sd, prefix + ".weight"), _get(sd, prefix + ".bias")
    mean, var = _get(sd, prefix + ".running_mean"), _get(sd, prefix + ".running_var")
    s = g / np.sqrt(var + BN_EPS)
    return w * s[:, None], (b - mean) * s + beta


def gin_weights_from_ogb_state_dict(sd: Mapping, num_layers: int = 5, eps_tol: float = 0.0, multi_task: bool = False) -> Dict[str, np.ndarray]:
    """OGB `GNN(gnn_type='gin', virtual_node=False, JK='last', residual=False, graph_pooling='mean')` -> the reference's
    GIN weight set.  `convs.l.mlp` may be Linear-BatchNorm-ReLU-Linear (OGB) or Linear-ReLU-Linear; `batch_norms.l`
    (applied to the conv output before the ReLU) is folded into the second linear layer when present.
    multi_task: keep every row of graph_pred_linear (ogbg-molpcba: 128 tasks) -- graph_pred_weights becomes [NUM_TASK][100], for
    an engine with flowgnn_set_num_tasks(NUM_TASK); without it a head with more than one task is refused (the reference's NUM_TASK is 1).
    The reference ignores GIN's eps (GIN/src/host_load.cc reads it, nothing uses it): a trained |eps| > eps_tol is refused."""
    p = "gnn_node"
    out = OrderedDict()
    out["node_embedding_weight"] = _tables(sd, f"{p}.atom_encoder.atom_embedding_list", ATOM_DIMS)
    ed, w1s, b1s, w2s, b2s = [], [], [], [], []
    for l in range(num_layers):
        c = f"{p}.convs.{l}"
        if f"{c}.eps" in sd and abs(float(_np(sd[f"{c}.eps"]).ravel()[0])) > eps_tol:
            raise ExportError(f"{c}.eps = {float(_np(sd[c + '.eps']).ravel()[0]):g}: the reference's GIN has no eps term")
        ed.append(_tables(sd, f"{c}.bond_encoder.bond_embedding_list", BOND_DIMS))
        w1, b1 = _get(sd, f"{c}.mlp.0.weight"), _get(sd, f"{c}.mlp.0.bias")
        if f"{c}.mlp.1.running_mean" in sd:
            w1, b1 = _fold_bn(w1, b1, sd, f"{c}.mlp.1")
            w2, b2 = _get(sd, f"{c}.mlp.3.weight"), _get(sd, f"{c}.mlp.3.bias")
        else:
            w2, b2 = _get(sd, f"{c}.mlp.2.weight"), _get(sd, f"{c}.mlp.2.bias")
        if f"{p}.batch_norms.{l}.running_mean" in sd:
            w2, b2 = _fold_bn(w2, b2, sd, f"{p}.batch_norms.{l}")
        w1s.append(w1); b1s.append(b1); w2s.append(w2); b2s.append(b2)
    out["edge_embedding_weight"] = np.stack(ed)
    out["node_mlp_1_weights"], out["node_mlp_1_bias"] = np.stack(w1s), np.stack(b1s)
    out["node_mlp_2_weights"], out["node_mlp_2_bias"] = np.stack(w2s), np.stack(b2s)
    pw, pb = _get(sd, "graph_pred_linear.weight"), _get(sd, "graph_pred_linear.bias")
    if pw.shape[0] != 1 and not multi_task:
        raise ExportError(f"graph_pred_linear has {pw.shape[0]} tasks; the reference is built with NUM_TASK = 1 (GIN/src/dcl.h:25): "
                          f"pass multi_task=True for an engine with flowgnn_set_num_tasks({pw.shape[0]})")
    out["graph_pred_weights"], out["graph_pred_bias"] = pw, pb
    return _checked(out, _weights.GIN_FILES, lambda k, v: _weights._task_shape(k, v[1], pw.shape[0]))


def gcn_weights_from_ogb_state_dict(sd: Mapping, num_layers: int = 5, multi_task: bool = False) -> Dict[str, np.ndarray]:
    """OGB `GNN(gnn_type='gcn', ...)` -> the reference's GCN weight set.  BatchNorm stays a separate step in the
    reference (GCN/src/node_embedding.cc:123-138) but with 2^-10 added to the variance instead of torch's 1e-5
    (GCN/src/load_inputs.cc:32): the exported variance is shifted by the difference so that both normalise alike."""
    p = "gnn_node"
    out = OrderedDict()
    out["node_embedding_weight"] = _tables(sd, f"{p}.atom_encoder.atom_embedding_list", ATOM_DIMS)
    cols = {k: [] for k in ("edge_embedding_weight", "convs_weight", "convs_bias", "convs_root_emb_weight", "bn_weight", "bn_bias",
                            "bn_mean", "bn_var")}
    for l in range(num_layers):
        c, bn = f"{p}.convs.{l}", f"{p}.batch_norms.{l}"
        cols["edge_embedding_weight"].append(_tables(sd, f"{c}.bond_encoder.bond_embedding_list", BOND_DIMS))
        cols["convs_weight"].append(_get(sd, f"{c}.linear.weight"))
        cols["convs_bias"].append(_get(sd, f"{c}.linear.bias"))
        cols["convs_root_emb_weight"].append(_get(sd, f"{c}.root_emb.weight").reshape(-1))
        cols["bn_weight"].append(_get(sd, f"{bn}.weight"))
        cols["bn_bias"].append(_get(sd, f"{bn}.bias"))
        cols["bn_mean"].append(_get(sd, f"{bn}.running_mean"))
        var = _get(sd, f"{bn}.running_var") + (BN_EPS - REF_BN_EPS)
        if (var + REF_BN_EPS <= 0).any():
            raise ExportError(f"{bn}.running_var is not positive")
        cols["bn_var"].append(var)
    for k, v in cols.items():
        out[k] = np.stack(v)
    pw, pb = _get(sd, "graph_pred_linear.weight"), _get(sd, "graph_pred_linear.bias")
    if pw.shape[0] != 1 and not multi_task:
        raise ExportError(f"graph_pred_linear has {pw.shape[0]} tasks; the reference is built with NUM_TASK = 1 (GCN/src/dcl.h): "
                          f"pass multi_task=True for an engine with flowgnn_set_num_tasks({pw.shape[0]})")
    out["graph_pred_weights"], out["graph_pred_bias"] = pw, pb
    ordered = OrderedDict((k, out[k]) for k in _weights.GCN_SHAPES)
    return _checked(ordered, _weights.GCN_SHAPES, lambda k, v: _weights._task_shape(k, v, pw.shape[0]))


def _checked(w: Dict[str, np.ndarray], spec: Mapping, shape_of) -> Dict[str, np.ndarray]:
    res = OrderedDict()
    for k, v in spec.items():
        shp = tuple(shape_of(k, v))
        a = np.asarray(w[k], dtype=np.float64)
        if a.size != int(np.prod(shp)):
            raise ExportError(f"{k}: {a.shape} does not fit the reference's {shp}")
        if not np.isfinite(a).all():
            raise ExportError(f"{k}: non-finite values")
        res[k] = np.ascontiguousarray(a.reshape(shp), dtype=np.float32)
    return res


def export_weights(model: str, sd: Mapping, directory: str, multi_task: bool = False) -> Dict[str, np.ndarray]:
    """Write the `.bin` file(s) `host` / `flowgnn_load_weights_dir` read for `model` ('GIN', 'GIN-VN' or 'GCN').
    multi_task: keep all rows of the prediction head (the engine then needs flowgnn_set_num_tasks(rows) before loading)."""
    m = model.upper()
    if m in ("GIN", "GIN-VN"):
        w = gin_weights_from_ogb_state_dict(sd, multi_task=multi_task)
        _weights.save_gin_weights(w, directory)
    elif m == "GCN":
        w = gcn_weights_from_ogb_state_dict(sd, multi_task=multi_task)
        _weights.save_gcn_weights(w, directory)
    else:
        raise ExportError(f"no OGB example model corresponds to the reference's {model} (its PNA/DGN/GAT checkpoints are already "
                          f"flat files: use flowgnn_amd.weights.save_*_weights)")
    return w


def _arr(g, name: str):
    v = g[name] if isinstance(g, Mapping) else getattr(g, name, None)
    if v is None:
        return None
    if hasattr(v, "detach"):
        v = v.detach().cpu().numpy()
    return np.asarray(v)


def batch_from_graphs(graphs: Iterable, with_eigen: bool = False) -> GraphBatch:
    """PyG-style graphs (`x` [N, 9] ints, `edge_index` [2, E] with both directions listed, `edge_attr` [E, 3] ints,
    optionally `eig` [N, >= 2] floats -- the DGN eigenvector file's columns) -> one GraphBatch with node ids local to each
    graph, the encoding of GIN/src/host_load.cc:100-143."""
    nn, ne, nf, el, ea, eg = [], [], [], [], [], []
    for i, g in enumerate(graphs):
        x, ei, at = _arr(g, "x"), _arr(g, "edge_index"), _arr(g, "edge_attr")
        if x is None or ei is None:
            raise ExportError(f"graph {i}: needs x and edge_index")
        if x.ndim != 2 or x.shape[1] != 9:
            raise ExportError(f"graph {i}: x is {x.shape}, the reference reads 9 integer features per node")
        if ei.ndim != 2 or ei.shape[0] != 2:
            raise ExportError(f"graph {i}: edge_index is {ei.shape}, wanted [2, E]")
        n, e = x.shape[0], ei.shape[1]
        if e and (ei.min() < 0 or ei.max() >= n):
            raise ExportError(f"graph {i}: edge_index refers to nodes outside [0, {n})")
        if at is None:
            at = np.zeros((e, 3), np.int64)
        if at.shape != (e, 3):
            raise ExportError(f"graph {i}: edge_attr is {at.shape}, wanted ({e}, 3)")
        nn.append(n); ne.append(e)
        nf.append(x.astype(np.int32)); el.append(ei.T.astype(np.int32)); ea.append(at.astype(np.int32))
        if with_eigen:
            v = _arr(g, "eig")
            if v is None or v.shape[0] != n or v.ndim != 2 or v.shape[1] < 2:
                raise ExportError(f"graph {i}: DGN needs eig [N, >= 2]")
            pad = np.zeros((n, 4), np.float32)
            pad[:, :min(4, v.shape[1])] = v[:, :4]
            eg.append(pad)
    cat = lambda xs, shp, dt: np.concatenate(xs).astype(dt) if xs else np.zeros(shp, dt)
    return GraphBatch(np.asarray(nn, np.int32), np.asarray(ne, np.int32), cat(nf, (0, 9), np.int32), cat(el, (0, 2), np.int32),
                      cat(ea, (0, 3), np.int32), cat(eg, (0, 4), np.float32) if with_eigen else None)


def export_dataset(graphs: Iterable, root: str, eig_dir: Optional[str] = None) -> GraphBatch:
    """Write the pack `host` reads: root/graph_info/g%d_info.txt, root/graph_bin/g%d_{node_feature,edge_list,edge_attr}.bin
    (1-based), root/dataset_size.txt, and eig_dir/g%d.txt for DGN."""
    b = batch_from_graphs(graphs, with_eigen=eig_dir is not None)
    write_pack(b, root, eig_dir=eig_dir)
    return b

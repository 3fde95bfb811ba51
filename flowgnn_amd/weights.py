"""Weight sets in the reference's raw little-endian float32 `.bin` layouts (no header),
plus seeded synthetic weights with the same shapes and scales.

Readers follow the reference host loaders:
  GIN   nine separate files                         GIN/src/host_load.cc:24-58
  GCN   one file, hard-coded float offsets          GCN/src/host_load.cc:31-170

Every weight set is an OrderedDict whose key order is the argument order of the model's
<M>_compute_graphs entry point (include/flowgnn.h).
"""
from __future__ import annotations

import os
from collections import OrderedDict
from typing import Dict

import numpy as np

# name -> (file, shape); order = argument order of flowgnn_set_weights_gin / GIN_compute_graphs
GIN_FILES = OrderedDict([
    ("node_embedding_weight", ("gin_ep1_nd_embed_dim100.bin", (173, 100))),
    ("edge_embedding_weight", ("gin_ep1_ed_embed_dim100.bin", (5, 13, 100))),
    ("node_mlp_1_weights", ("gin_ep1_mlp_1_weights_dim100.bin", (5, 200, 100))),
    ("node_mlp_1_bias", ("gin_ep1_mlp_1_bias_dim100.bin", (5, 200))),
    ("node_mlp_2_weights", ("gin_ep1_mlp_2_weights_dim100.bin", (5, 100, 200))),
    ("node_mlp_2_bias", ("gin_ep1_mlp_2_bias_dim100.bin", (5, 100))),
    ("graph_pred_weights", ("gin_ep1_pred_weights_dim100.bin", (1, 100))),
    ("graph_pred_bias", ("gin_ep1_pred_bias_dim100.bin", (1,))),
])


def _read(path: str, shape, offset_floats: int = 0) -> np.ndarray:
    count = int(np.prod(shape))
    arr = np.fromfile(path, dtype="<f4", count=count, offset=4 * offset_floats)
    if arr.size != count:
        raise IOError(f"{path}: wanted {count} floats at offset {offset_floats}, got {arr.size}")
    return np.ascontiguousarray(arr.reshape(shape), dtype=np.float32)


def load_gin_weights(directory: str) -> Dict[str, np.ndarray]:
    return OrderedDict((k, _read(os.path.join(directory, f), shp)) for k, (f, shp) in GIN_FILES.items())


def save_gin_weights(w: Dict[str, np.ndarray], directory: str) -> None:
    os.makedirs(directory, exist_ok=True)
    for k, (f, shp) in GIN_FILES.items():
        np.asarray(w[k], dtype="<f4").reshape(shp).tofile(os.path.join(directory, f))
    np.zeros(5, dtype="<f4").tofile(os.path.join(directory, "gin_ep1_eps_dim100.bin"))  # read, never used


def synth_gin_weights(seed: int = 7) -> Dict[str, np.ndarray]:
    """Random weights with the measured scales of the shipped GIN set:
    W1 s=0.075, b1 s=0.91, W2 s=0.053, b2 s=0.49, node/edge emb s=0.11/0.14, pred s=0.15."""
    rng = np.random.default_rng(seed)
    scale = {
        "node_embedding_weight": 0.11, "edge_embedding_weight": 0.14,
        "node_mlp_1_weights": 0.075, "node_mlp_1_bias": 0.91,
        "node_mlp_2_weights": 0.053, "node_mlp_2_bias": 0.49,
        "graph_pred_weights": 0.15, "graph_pred_bias": 0.12,
    }
    return OrderedDict((k, (rng.standard_normal(shp) * scale[k]).astype(np.float32))
                       for k, (_, shp) in GIN_FILES.items())


# --------------------------------------------------------------------------- GCN
GCN_FILE = "gcn_ep1_dim100.weights.all.bin"
GCN_SHAPES = OrderedDict([
    ("node_embedding_weight", (173, 100)), ("edge_embedding_weight", (5, 13, 100)),
    ("convs_weight", (5, 100, 100)), ("convs_bias", (5, 100)), ("convs_root_emb_weight", (5, 100)),
    ("bn_weight", (5, 100)), ("bn_bias", (5, 100)), ("bn_mean", (5, 100)), ("bn_var", (5, 100)),
    ("graph_pred_weights", (1, 100)), ("graph_pred_bias", (1,)),
])


def _gcn_offsets():
    """(name, layer or None) -> float offset in the .all.bin (GCN/src/host_load.cc:34-170)."""
    off = {("node_embedding_weight", None): 0, ("graph_pred_weights", None): 76805, ("graph_pred_bias", None): 76905}
    for l in range(5):
        base = 17300 + 11500 * l
        off[("convs_weight", l)] = base
        off[("convs_bias", l)] = base + 10000
        off[("convs_root_emb_weight", l)] = base + 10100
        off[("edge_embedding_weight", l)] = base + 10200
        bn = 74800 + 401 * l  # 4 x 100 floats, then one skipped counter (num_batches_tracked)
        off[("bn_weight", l)] = bn
        off[("bn_bias", l)] = bn + 100
        off[("bn_mean", l)] = bn + 200
        off[("bn_var", l)] = bn + 300
    return off


def load_gcn_weights(directory: str) -> Dict[str, np.ndarray]:
    path = os.path.join(directory, GCN_FILE)
    off = _gcn_offsets()
    w = OrderedDict()
    for k, shp in GCN_SHAPES.items():
        if (k, None) in off:
            w[k] = _read(path, shp, off[(k, None)])
        else:
            w[k] = np.stack([_read(path, shp[1:], off[(k, l)]) for l in range(5)])
    return w


def save_gcn_weights(w: Dict[str, np.ndarray], directory: str) -> None:
    os.makedirs(directory, exist_ok=True)
    buf = np.zeros(76906, dtype="<f4")
    off = _gcn_offsets()
    for k, shp in GCN_SHAPES.items():
        a = np.asarray(w[k], dtype=np.float32).reshape(shp)
        if (k, None) in off:
            buf[off[(k, None)]:off[(k, None)] + a.size] = a.ravel()
        else:
            for l in range(5):
                buf[off[(k, l)]:off[(k, l)] + a[l].size] = a[l].ravel()
    buf.tofile(os.path.join(directory, GCN_FILE))


def synth_gcn_weights(seed: int = 7) -> Dict[str, np.ndarray]:
    rng = np.random.default_rng(seed)
    scale = {"node_embedding_weight": 0.11, "edge_embedding_weight": 0.14, "convs_weight": 0.09, "convs_bias": 0.1,
             "convs_root_emb_weight": 0.3, "bn_weight": 0.2, "bn_bias": 0.2, "bn_mean": 0.3,
             "graph_pred_weights": 0.15, "graph_pred_bias": 0.12}
    w = OrderedDict()
    for k, shp in GCN_SHAPES.items():
        if k == "bn_var":
            w[k] = rng.uniform(0.3, 1.5, shp).astype(np.float32)
        elif k == "bn_weight":
            w[k] = (1.0 + rng.standard_normal(shp) * scale[k]).astype(np.float32)
        else:
            w[k] = (rng.standard_normal(shp) * scale[k]).astype(np.float32)
    return w


LOADERS = {"GIN": load_gin_weights, "GIN-VN": load_gin_weights, "GCN": load_gcn_weights}
SYNTH = {"GIN": synth_gin_weights, "GIN-VN": synth_gin_weights, "GCN": synth_gcn_weights}
SAVERS = {"GIN": save_gin_weights, "GIN-VN": save_gin_weights, "GCN": save_gcn_weights}

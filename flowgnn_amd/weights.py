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


def _task_shape(k: str, shp, num_tasks: int):
    """graph_pred_weights [NUM_TASK][D] / graph_pred_bias [NUM_TASK]: NUM_TASK is 1 in the reference (GIN/src/dcl.h:25), a
    run-time dimension here (flowgnn_set_num_tasks)."""
    if k == "graph_pred_weights":
        return (num_tasks,) + tuple(shp[1:])
    if k == "graph_pred_bias":
        return (num_tasks,)
    return tuple(shp)


def load_gin_weights(directory: str, num_tasks: int = 1) -> Dict[str, np.ndarray]:
    return OrderedDict((k, _read(os.path.join(directory, f), _task_shape(k, shp, num_tasks))) for k, (f, shp) in GIN_FILES.items())


def save_gin_weights(w: Dict[str, np.ndarray], directory: str) -> None:
    os.makedirs(directory, exist_ok=True)
    tasks = int(np.asarray(w["graph_pred_bias"]).size)
    for k, (f, shp) in GIN_FILES.items():
        np.asarray(w[k], dtype="<f4").reshape(_task_shape(k, shp, tasks)).tofile(os.path.join(directory, f))
    np.zeros(5, dtype="<f4").tofile(os.path.join(directory, "gin_ep1_eps_dim100.bin"))  # read, never used


def synth_gin_weights(seed: int = 7, num_tasks: int = 1) -> Dict[str, np.ndarray]:
    """Random weights with the measured scales of the shipped GIN set:
    W1 s=0.075, b1 s=0.91, W2 s=0.053, b2 s=0.49, node/edge emb s=0.11/0.14, pred s=0.15."""
    rng = np.random.default_rng(seed)
    scale = {
        "node_embedding_weight": 0.11, "edge_embedding_weight": 0.14,
        "node_mlp_1_weights": 0.075, "node_mlp_1_bias": 0.91,
        "node_mlp_2_weights": 0.053, "node_mlp_2_bias": 0.49,
        "graph_pred_weights": 0.15, "graph_pred_bias": 0.12,
    }
    return OrderedDict((k, (rng.standard_normal(_task_shape(k, shp, num_tasks)) * scale[k]).astype(np.float32))
                       for k, (_, shp) in GIN_FILES.items())


# --------------------------------------------------------------------------- GCN
GCN_FILE = "gcn_ep1_dim100.weights.all.bin"
GCN_SHAPES = OrderedDict([
    ("node_embedding_weight", (173, 100)), ("edge_embedding_weight", (5, 13, 100)),
    ("convs_weight", (5, 100, 100)), ("convs_bias", (5, 100)), ("convs_root_emb_weight", (5, 100)),
    ("bn_weight", (5, 100)), ("bn_bias", (5, 100)), ("bn_mean", (5, 100)), ("bn_var", (5, 100)),
    ("graph_pred_weights", (1, 100)), ("graph_pred_bias", (1,)),
])


def _gcn_offsets(num_tasks: int = 1):
    """(name, layer or None) -> float offset in the .all.bin (GCN/src/host_load.cc:34-170); the head is the last tensor pair of
    the flattened state_dict: weight [NUM_TASK][100] at 76805, bias [NUM_TASK] right behind it (76905 for NUM_TASK = 1)."""
    off = {("node_embedding_weight", None): 0, ("graph_pred_weights", None): 76805, ("graph_pred_bias", None): 76805 + 100 * num_tasks}
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


def load_gcn_weights(directory: str, num_tasks: int = 1) -> Dict[str, np.ndarray]:
    path = os.path.join(directory, GCN_FILE)
    off = _gcn_offsets(num_tasks)
    w = OrderedDict()
    for k, shp in GCN_SHAPES.items():
        if (k, None) in off:
            w[k] = _read(path, _task_shape(k, shp, num_tasks), off[(k, None)])
        else:
            w[k] = np.stack([_read(path, shp[1:], off[(k, l)]) for l in range(5)])
    return w


def save_gcn_weights(w: Dict[str, np.ndarray], directory: str) -> None:
    os.makedirs(directory, exist_ok=True)
    tasks = int(np.asarray(w["graph_pred_bias"]).size)
    buf = np.zeros(76805 + 101 * tasks, dtype="<f4")
    off = _gcn_offsets(tasks)
    for k, shp in GCN_SHAPES.items():
        a = np.asarray(w[k], dtype=np.float32).reshape(_task_shape(k, shp, tasks))
        if (k, None) in off:
            buf[off[(k, None)]:off[(k, None)] + a.size] = a.ravel()
        else:
            for l in range(5):
                buf[off[(k, l)]:off[(k, l)] + a[l].size] = a[l].ravel()
    buf.tofile(os.path.join(directory, GCN_FILE))


def synth_gcn_weights(seed: int = 7, num_tasks: int = 1) -> Dict[str, np.ndarray]:
    rng = np.random.default_rng(seed)
    scale = {"node_embedding_weight": 0.11, "edge_embedding_weight": 0.14, "convs_weight": 0.09, "convs_bias": 0.1,
             "convs_root_emb_weight": 0.3, "bn_weight": 0.2, "bn_bias": 0.2, "bn_mean": 0.3,
             "graph_pred_weights": 0.15, "graph_pred_bias": 0.12}
    w = OrderedDict()
    for k, shp in GCN_SHAPES.items():
        shp = _task_shape(k, shp, num_tasks)
        if k == "bn_var":
            w[k] = rng.uniform(0.3, 1.5, shp).astype(np.float32)
        elif k == "bn_weight":
            w[k] = (1.0 + rng.standard_normal(shp) * scale[k]).astype(np.float32)
        else:
            w[k] = (rng.standard_normal(shp) * scale[k]).astype(np.float32)
    return w


# --------------------------------------------------------------------------- PNA
PNA_FILE = "pna_ep1_noBN_dim80.weights.all.bin"
PNA_AVG_DEG = 6.885701656341553  # PNA/src/host_load.cc:127 (host constant, not in the file)
PNA_SHAPES = OrderedDict([
    ("node_embedding_weight", (173, 80)), ("node_conv_weights", (4, 80, 3, 4, 80)), ("node_conv_bias", (4, 80)),
    ("graph_mlp_1_weights", (40, 80)), ("graph_mlp_1_bias", (40,)), ("graph_mlp_2_weights", (20, 40)),
    ("graph_mlp_2_bias", (20,)), ("graph_mlp_3_weights", (1, 20)), ("graph_mlp_3_bias", (1,)), ("avg_deg", (1,)),
])
_PNA_TAIL = OrderedDict([("graph_mlp_1_weights", 321360), ("graph_mlp_1_bias", 324560), ("graph_mlp_2_weights", 324600),
                         ("graph_mlp_2_bias", 325400), ("graph_mlp_3_weights", 325420), ("graph_mlp_3_bias", 325440)])


def load_pna_weights(directory: str) -> Dict[str, np.ndarray]:
    """PNA/src/host_load.cc:23-68: node_emb at 0, layer l weights at 13840 + 76880 l (76800) + bias (80)."""
    path = os.path.join(directory, PNA_FILE)
    w = OrderedDict()
    w["node_embedding_weight"] = _read(path, (173, 80), 0)
    w["node_conv_weights"] = np.stack([_read(path, (80, 3, 4, 80), 13840 + 76880 * l) for l in range(4)])
    w["node_conv_bias"] = np.stack([_read(path, (80,), 13840 + 76880 * l + 76800) for l in range(4)])
    for k, off in _PNA_TAIL.items():
        w[k] = _read(path, PNA_SHAPES[k], off)
    w["avg_deg"] = np.array([PNA_AVG_DEG], dtype=np.float32)
    return w


def save_pna_weights(w: Dict[str, np.ndarray], directory: str) -> None:
    os.makedirs(directory, exist_ok=True)
    buf = np.zeros(325441, dtype="<f4")
    buf[0:13840] = np.asarray(w["node_embedding_weight"], np.float32).ravel()
    for l in range(4):
        base = 13840 + 76880 * l
        buf[base:base + 76800] = np.asarray(w["node_conv_weights"], np.float32)[l].ravel()
        buf[base + 76800:base + 76880] = np.asarray(w["node_conv_bias"], np.float32)[l].ravel()
    for k, off in _PNA_TAIL.items():
        a = np.asarray(w[k], np.float32).ravel()
        buf[off:off + a.size] = a
    buf.tofile(os.path.join(directory, PNA_FILE))


def synth_pna_weights(seed: int = 7) -> Dict[str, np.ndarray]:
    rng = np.random.default_rng(seed)
    scale = {"node_embedding_weight": 0.11, "node_conv_weights": 0.012, "node_conv_bias": 0.1, "graph_mlp_1_weights": 0.1,
             "graph_mlp_1_bias": 0.1, "graph_mlp_2_weights": 0.15, "graph_mlp_2_bias": 0.1, "graph_mlp_3_weights": 0.2,
             "graph_mlp_3_bias": 0.1}
    w = OrderedDict((k, (rng.standard_normal(shp) * scale[k]).astype(np.float32)) for k, shp in PNA_SHAPES.items()
                    if k != "avg_deg")
    w["avg_deg"] = np.array([PNA_AVG_DEG], dtype=np.float32)
    return w


# --------------------------------------------------------------------------- DGN
DGN_FILE = "dgn_ep1_noBN_dim100.weights.all.bin"
DGN_SHAPES = OrderedDict([
    ("embedding_h_atom_embedding_list_weights", (9, 119, 100)),
    ("layers_posttrans_fully_connected_0_linear_weight", (4, 100, 200)),
    ("layers_posttrans_fully_connected_0_linear_bias", (4, 100)),
    ("MLP_layer_FC_layers_0_weight", (50, 100)), ("MLP_layer_FC_layers_0_bias", (50,)),
    ("MLP_layer_FC_layers_1_weight", (25, 50)), ("MLP_layer_FC_layers_1_bias", (25,)),
    ("MLP_layer_FC_layers_2_weight", (1, 25)), ("MLP_layer_FC_layers_2_bias", (1,)),
])
_DGN_TABLE_OFF = [0, 11900, 12300, 13500, 14700, 15700, 16300, 16900, 17100]  # DGN/src/host_load.cc:12-64
_DGN_TAIL = OrderedDict([("MLP_layer_FC_layers_0_weight", 97700), ("MLP_layer_FC_layers_0_bias", 102700),
                         ("MLP_layer_FC_layers_1_weight", 102750), ("MLP_layer_FC_layers_1_bias", 104000),
                         ("MLP_layer_FC_layers_2_weight", 104025), ("MLP_layer_FC_layers_2_bias", 104050)])
_CARD = [119, 4, 12, 12, 10, 6, 6, 2, 2]


def load_dgn_weights(directory: str) -> Dict[str, np.ndarray]:
    path = os.path.join(directory, DGN_FILE)
    emb = np.zeros((9, 119, 100), dtype=np.float32)
    for k in range(9):
        emb[k, :_CARD[k]] = _read(path, (_CARD[k], 100), _DGN_TABLE_OFF[k])
    w = OrderedDict()
    w["embedding_h_atom_embedding_list_weights"] = emb
    w["layers_posttrans_fully_connected_0_linear_weight"] = np.stack(
        [_read(path, (100, 200), 17300 + 20100 * l) for l in range(4)])
    w["layers_posttrans_fully_connected_0_linear_bias"] = np.stack(
        [_read(path, (100,), 17300 + 20100 * l + 20000) for l in range(4)])
    for k, off in _DGN_TAIL.items():
        w[k] = _read(path, DGN_SHAPES[k], off)
    return w


def save_dgn_weights(w: Dict[str, np.ndarray], directory: str) -> None:
    os.makedirs(directory, exist_ok=True)
    buf = np.zeros(104051, dtype="<f4")
    emb = np.asarray(w["embedding_h_atom_embedding_list_weights"], np.float32)
    for k in range(9):
        buf[_DGN_TABLE_OFF[k]:_DGN_TABLE_OFF[k] + _CARD[k] * 100] = emb[k, :_CARD[k]].ravel()
    for l in range(4):
        base = 17300 + 20100 * l
        buf[base:base + 20000] = np.asarray(w["layers_posttrans_fully_connected_0_linear_weight"], np.float32)[l].ravel()
        buf[base + 20000:base + 20100] = np.asarray(w["layers_posttrans_fully_connected_0_linear_bias"], np.float32)[l].ravel()
    for k, off in _DGN_TAIL.items():
        a = np.asarray(w[k], np.float32).ravel()
        buf[off:off + a.size] = a
    buf.tofile(os.path.join(directory, DGN_FILE))


def synth_dgn_weights(seed: int = 7) -> Dict[str, np.ndarray]:
    rng = np.random.default_rng(seed)
    scale = {"embedding_h_atom_embedding_list_weights": 0.11, "layers_posttrans_fully_connected_0_linear_weight": 0.05,
             "layers_posttrans_fully_connected_0_linear_bias": 0.1, "MLP_layer_FC_layers_0_weight": 0.1,
             "MLP_layer_FC_layers_0_bias": 0.1, "MLP_layer_FC_layers_1_weight": 0.15, "MLP_layer_FC_layers_1_bias": 0.1,
             "MLP_layer_FC_layers_2_weight": 0.2, "MLP_layer_FC_layers_2_bias": 0.1}
    w = OrderedDict((k, (rng.standard_normal(shp) * scale[k]).astype(np.float32)) for k, shp in DGN_SHAPES.items())
    for k in range(9):  # rows beyond a table's cardinality do not exist in the file
        w["embedding_h_atom_embedding_list_weights"][k, _CARD[k]:] = 0.0
    return w


# --------------------------------------------------------------------------- GAT
GAT_SHAPES = OrderedDict([
    ("scoring_fn_target", (5, 4, 16)), ("scoring_fn_source", (5, 4, 16)),
    ("linear_proj_weights", (5, 4, 16, 4, 16)), ("skip_proj_weights", (5, 4, 16, 4, 16)),
    ("graph_pred_weights", (1, 16)), ("graph_pred_bias", (1,)),
])
_GAT_FILES = {"graph_pred_weights": "gat_ep1_pred_weights_layer5.bin", "graph_pred_bias": "gat_ep1_pred_bias_layer5.bin",
              "scoring_fn_target": "gat_ep1_scoring_fn_target_layer5.bin",
              "scoring_fn_source": "gat_ep1_scoring_fn_source_layer5.bin"}


def load_gat_weights(directory: str) -> Dict[str, np.ndarray]:
    """GAT/src/host_load.cc:20-91: layer 0 of linear/skip proj is a [4][16][1][9] file placed in the
    [head_out][dim_out][head_in = 0][dim_in < 9] corner of a zero [4][16][4][16] tensor."""
    w = OrderedDict()
    for k in ("scoring_fn_target", "scoring_fn_source"):
        w[k] = _read(os.path.join(directory, _GAT_FILES[k]), GAT_SHAPES[k])
    for k, stem in (("linear_proj_weights", "linear_proj_weight"), ("skip_proj_weights", "skip_proj_weight")):
        full = np.zeros((5, 4, 16, 4, 16), dtype=np.float32)
        full[0, :, :, 0, :9] = _read(os.path.join(directory, f"gat_ep1_{stem}_0_layer5.bin"), (4, 16, 9))
        full[1:] = _read(os.path.join(directory, f"gat_ep1_{stem}_1_layer5.bin"), (4, 4, 16, 4, 16))
        w[k] = full
    for k in ("graph_pred_weights", "graph_pred_bias"):
        w[k] = _read(os.path.join(directory, _GAT_FILES[k]), GAT_SHAPES[k])
    return w


def save_gat_weights(w: Dict[str, np.ndarray], directory: str) -> None:
    os.makedirs(directory, exist_ok=True)
    for k, f in _GAT_FILES.items():
        np.asarray(w[k], dtype="<f4").tofile(os.path.join(directory, f))
    for k, stem in (("linear_proj_weights", "linear_proj_weight"), ("skip_proj_weights", "skip_proj_weight")):
        a = np.asarray(w[k], dtype="<f4")
        np.ascontiguousarray(a[0, :, :, 0, :9]).tofile(os.path.join(directory, f"gat_ep1_{stem}_0_layer5.bin"))
        np.ascontiguousarray(a[1:]).tofile(os.path.join(directory, f"gat_ep1_{stem}_1_layer5.bin"))


def synth_gat_weights(seed: int = 7) -> Dict[str, np.ndarray]:
    """Small weights: the reference feeds raw atom numbers (0..118) into layer 0 and exponentiates the
    attention scores without max subtraction, so the scale has to keep exp() finite."""
    rng = np.random.default_rng(seed)
    scale = {"scoring_fn_target": 0.1, "scoring_fn_source": 0.1, "linear_proj_weights": 0.08, "skip_proj_weights": 0.08,
             "graph_pred_weights": 0.2, "graph_pred_bias": 0.1}
    w = OrderedDict((k, (rng.standard_normal(shp) * scale[k]).astype(np.float32)) for k, shp in GAT_SHAPES.items())
    for k in ("linear_proj_weights", "skip_proj_weights"):
        corner = w[k][0, :, :, 0, :9].copy() * 0.02  # raw integer inputs
        w[k][0] = 0.0
        w[k][0, :, :, 0, :9] = corner
    return w


LOADERS = {"GIN": load_gin_weights, "GIN-VN": load_gin_weights, "GCN": load_gcn_weights, "PNA": load_pna_weights, "DGN": load_dgn_weights, "GAT": load_gat_weights}
SYNTH = {"GIN": synth_gin_weights, "GIN-VN": synth_gin_weights, "GCN": synth_gcn_weights, "PNA": synth_pna_weights, "DGN": synth_dgn_weights, "GAT": synth_gat_weights}
SAVERS = {"GIN": save_gin_weights, "GIN-VN": save_gin_weights, "GCN": save_gcn_weights, "PNA": save_pna_weights, "DGN": save_dgn_weights, "GAT": save_gat_weights}

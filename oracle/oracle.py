"""ctypes binding of oracle/liboracle.so -- TEST INFRASTRUCTURE ONLY (see flowgnn_oracle.h).

Importable only from tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "liboracle.so")
_lib = None

_pi = C.POINTER(C.c_int)
_pf = C.POINTER(C.c_float)


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE])


def load():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB):
            build()
        lib = C.CDLL(_LIB)
        lib.orc_gin_load_graph.argtypes = [_pi, _pi, C.c_int, C.c_int, _pi, _pi, _pi, _pi, _pi]
        lib.orc_gin_load_graph.restype = None
        lib.orc_GIN_compute_graphs.argtypes = [C.c_int, _pi, _pi, _pi, _pf, _pi, _pi, _pi] + [_pf] * 8 + [_pf, C.c_int]
        lib.orc_GIN_compute_graphs.restype = C.c_int
        lib.orc_GCN_compute_graphs.argtypes = [C.c_int, _pi, _pi, _pi, _pf, _pi, _pi, _pi] + [_pf] * 11 + [_pf, C.c_int]
        lib.orc_GCN_compute_graphs.restype = C.c_int
        lib.orc_PNA_compute_graphs.argtypes = [C.c_int, _pi, _pi, _pi, _pf, _pi, _pi] + [_pf] * 10 + [_pf, C.c_int]
        lib.orc_PNA_compute_graphs.restype = C.c_int
        lib.orc_DGN_compute_graphs.argtypes = [C.c_int, _pi, _pi, _pi, _pf, _pi, _pf, _pi] + [_pf] * 9 + [_pf, C.c_int]
        lib.orc_DGN_compute_graphs.restype = C.c_int
        lib.orc_GAT_compute_graphs.argtypes = [C.c_int, _pi, _pi, _pi, _pf, _pi, _pi] + [_pf] * 6 + [C.c_int, _pf, C.c_int]
        lib.orc_GAT_compute_graphs.restype = C.c_int
        _lib = lib
    return _lib


def _i(a):
    return np.ascontiguousarray(a, dtype=np.int32)


def _f(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def gin_load_graph(edge_list, edge_attr, n):
    """Reference load_graph tables of one graph (GIN/src/load_inputs.cc:87-172)."""
    lib = load()
    el, ea = _i(edge_list).reshape(-1, 2), _i(edge_attr).reshape(-1, 3)
    e = el.shape[0]
    deg = np.zeros(max(n, 1), np.int32)
    degs = np.zeros(4 * max(n, 1), np.int32)
    nbr = np.zeros(4 * max(e, 1), np.int32)
    att = np.zeros(4 * max(e, 1) * 3, np.int32)
    epp = np.zeros(4, np.int32)
    lib.orc_gin_load_graph(el.ctypes.data_as(_pi), ea.ctypes.data_as(_pi), n, e, deg.ctypes.data_as(_pi),
                           degs.ctypes.data_as(_pi), nbr.ctypes.data_as(_pi), att.ctypes.data_as(_pi),
                           epp.ctypes.data_as(_pi))
    return dict(degree_table=deg[:n], degree_tables=degs.reshape(4, -1)[:, :n],
                neighbor_tables=nbr.reshape(4, -1), edge_attrs=att.reshape(4, -1, 3), num_of_edges_per_pe=epp)


def gin_forward(batch, weight_sets, reload_weights=None, dump_h=False, nthreads=1, num_tasks=1):
    """orc_GIN_compute_graphs over a GraphBatch; weight_sets = list of weight dicts.  num_tasks > 1: graph_pred_weights
    [num_tasks][100], graph_pred_bias [num_tasks], result [G][num_tasks] (orc_GIN_compute_graphs_mt)."""
    lib = load()
    G = batch.num_graphs
    if reload_weights is None:
        reload_weights = np.zeros(G, np.int32)
        if G:
            reload_weights[0] = 1
    keys = list(weight_sets[0].keys())
    stacked = [_f(np.stack([np.asarray(ws[k], np.float32) for ws in weight_sets])) for k in keys]
    out = np.zeros(G * num_tasks, np.float32)
    nn, ne, rw = _i(batch.nums_of_nodes), _i(batch.nums_of_edges), _i(reload_weights)
    nf, el, ea = _i(batch.node_feature), _i(batch.edge_list), _i(batch.edge_attr)
    hd = np.zeros((6, batch.total_nodes, 100), np.float32) if dump_h else None
    lib.orc_GIN_compute_graphs_mt.argtypes = [C.c_int, _pi, _pi, _pi, _pf, _pi, _pi, _pi] + [_pf] * 8 + [_pf, C.c_int, C.c_int]
    lib.orc_GIN_compute_graphs_mt.restype = C.c_int
    rc = lib.orc_GIN_compute_graphs_mt(G, nn.ctypes.data_as(_pi), ne.ctypes.data_as(_pi), rw.ctypes.data_as(_pi),
                                       out.ctypes.data_as(_pf), nf.ctypes.data_as(_pi), el.ctypes.data_as(_pi),
                                       ea.ctypes.data_as(_pi), *[a.ctypes.data_as(_pf) for a in stacked],
                                       None if hd is None else hd.ctypes.data_as(_pf), nthreads, num_tasks)
    if rc:
        raise RuntimeError(f"oracle GIN rc={rc}")
    if num_tasks > 1:
        out = out.reshape(G, num_tasks)
    return (out, hd) if dump_h else out


def gin_forward_q(batch, weight_sets, reload_weights=None, dump_h=False, nthreads=1):
    """orc_GIN_compute_graphs_q: GIN in ap_fixed<16,6> (Q6.10).  Returns (logits as float, 16-bit patterns[, h dump int16])."""
    lib = load()
    G = batch.num_graphs
    if reload_weights is None:
        reload_weights = np.zeros(G, np.int32)
        if G:
            reload_weights[0] = 1
    keys = list(weight_sets[0].keys())
    stacked = [_f(np.stack([np.asarray(ws[k], np.float32) for ws in weight_sets])) for k in keys]
    out = np.zeros(G, np.float32)
    out_q = np.zeros(G, np.int16)
    nn, ne, rw = _i(batch.nums_of_nodes), _i(batch.nums_of_edges), _i(reload_weights)
    nf, el, ea = _i(batch.node_feature), _i(batch.edge_list), _i(batch.edge_attr)
    hd = np.zeros((6, batch.total_nodes, 100), np.int16) if dump_h else None
    p16 = C.POINTER(C.c_int16)
    lib.orc_GIN_compute_graphs_q.restype = C.c_int
    rc = lib.orc_GIN_compute_graphs_q(G, nn.ctypes.data_as(_pi), ne.ctypes.data_as(_pi), rw.ctypes.data_as(_pi),
                                      out.ctypes.data_as(_pf), out_q.ctypes.data_as(p16), nf.ctypes.data_as(_pi),
                                      el.ctypes.data_as(_pi), ea.ctypes.data_as(_pi), *[a.ctypes.data_as(_pf) for a in stacked],
                                      None if hd is None else hd.ctypes.data_as(p16), nthreads)
    if rc:
        raise RuntimeError(f"oracle GIN-Q rc={rc}")
    return (out, out_q, hd) if dump_h else (out, out_q)


def _forward(fn_name, batch, weight_sets, reload_weights, dump_shape, nthreads, with_attr=True, eig=False):
    lib = load()
    G = batch.num_graphs
    if reload_weights is None:
        reload_weights = np.zeros(G, np.int32)
        if G:
            reload_weights[0] = 1
    keys = list(weight_sets[0].keys())
    stacked = [_f(np.stack([np.asarray(ws[k], np.float32) for ws in weight_sets])) for k in keys]
    out = np.zeros(G, np.float32)
    nn, ne, rw = _i(batch.nums_of_nodes), _i(batch.nums_of_edges), _i(reload_weights)
    nf, el, ea = _i(batch.node_feature), _i(batch.edge_list), _i(batch.edge_attr)
    hd = np.zeros(dump_shape, np.float32) if dump_shape is not None else None
    args = [G, nn.ctypes.data_as(_pi), ne.ctypes.data_as(_pi), rw.ctypes.data_as(_pi), out.ctypes.data_as(_pf),
            nf.ctypes.data_as(_pi)]
    if eig:
        ev = _f(batch.node_eigen)
        args.append(ev.ctypes.data_as(_pf))
    args.append(el.ctypes.data_as(_pi))
    if with_attr:
        args.append(ea.ctypes.data_as(_pi))
    args += [a.ctypes.data_as(_pf) for a in stacked]
    args += [None if hd is None else hd.ctypes.data_as(_pf), nthreads]
    rc = getattr(lib, fn_name)(*args)
    if rc:
        raise RuntimeError(f"oracle {fn_name} rc={rc}")
    return (out, hd) if hd is not None else out


def gcn_forward(batch, weight_sets, reload_weights=None, dump_h=False, nthreads=1, num_tasks=1):
    """orc_GCN_compute_graphs; dump = x_l (NT outputs) [5][N][100].  num_tasks > 1: result [G][num_tasks]."""
    if num_tasks == 1:
        return _forward("orc_GCN_compute_graphs", batch, weight_sets, reload_weights,
                        (5, batch.total_nodes, 100) if dump_h else None, nthreads)
    lib = load()
    lib.orc_GCN_compute_graphs_mt.argtypes = [C.c_int, _pi, _pi, _pi, _pf, _pi, _pi, _pi] + [_pf] * 11 + [_pf, C.c_int, C.c_int]
    lib.orc_GCN_compute_graphs_mt.restype = C.c_int
    G = batch.num_graphs
    if reload_weights is None:
        reload_weights = np.zeros(G, np.int32)
        if G:
            reload_weights[0] = 1
    keys = list(weight_sets[0].keys())
    stacked = [_f(np.stack([np.asarray(ws[k], np.float32) for ws in weight_sets])) for k in keys]
    out = np.zeros(G * num_tasks, np.float32)
    nn, ne, rw = _i(batch.nums_of_nodes), _i(batch.nums_of_edges), _i(reload_weights)
    nf, el, ea = _i(batch.node_feature), _i(batch.edge_list), _i(batch.edge_attr)
    hd = np.zeros((5, batch.total_nodes, 100), np.float32) if dump_h else None
    rc = lib.orc_GCN_compute_graphs_mt(G, nn.ctypes.data_as(_pi), ne.ctypes.data_as(_pi), rw.ctypes.data_as(_pi),
                                       out.ctypes.data_as(_pf), nf.ctypes.data_as(_pi), el.ctypes.data_as(_pi),
                                       ea.ctypes.data_as(_pi), *[a.ctypes.data_as(_pf) for a in stacked],
                                       None if hd is None else hd.ctypes.data_as(_pf), nthreads, num_tasks)
    if rc:
        raise RuntimeError(f"oracle GCN rc={rc}")
    out = out.reshape(G, num_tasks)
    return (out, hd) if dump_h else out


def pna_forward(batch, weight_sets, reload_weights=None, dump_h=False, nthreads=1):
    """orc_PNA_compute_graphs; dump = h after encoder and each layer, [5][N][80]."""
    return _forward("orc_PNA_compute_graphs", batch, weight_sets, reload_weights,
                    (5, batch.total_nodes, 80) if dump_h else None, nthreads, with_attr=False)


def dgn_forward(batch, weight_sets, reload_weights=None, dump_h=False, nthreads=1):
    """orc_DGN_compute_graphs; dump = h after encoder and each layer, [5][N][100]."""
    return _forward("orc_DGN_compute_graphs", batch, weight_sets, reload_weights,
                    (5, batch.total_nodes, 100) if dump_h else None, nthreads, with_attr=False, eig=True)


def gat_forward(batch, weight_sets, reload_weights=None, dump_h=False, nthreads=1, feature_offset_quirk=False):
    """orc_GAT_compute_graphs; dump = ELU outputs of layers 0..3, [4][N][64] (index dim*4 + head)."""
    lib = load()
    G = batch.num_graphs
    if reload_weights is None:
        reload_weights = np.zeros(G, np.int32)
        if G:
            reload_weights[0] = 1
    keys = list(weight_sets[0].keys())
    stacked = [_f(np.stack([np.asarray(ws[k], np.float32) for ws in weight_sets])) for k in keys]
    out = np.zeros(G, np.float32)
    nn, ne, rw = _i(batch.nums_of_nodes), _i(batch.nums_of_edges), _i(reload_weights)
    nf, el = _i(batch.node_feature), _i(batch.edge_list)
    hd = np.zeros((4, batch.total_nodes, 64), np.float32) if dump_h else None
    rc = lib.orc_GAT_compute_graphs(G, nn.ctypes.data_as(_pi), ne.ctypes.data_as(_pi), rw.ctypes.data_as(_pi),
                                    out.ctypes.data_as(_pf), nf.ctypes.data_as(_pi), el.ctypes.data_as(_pi),
                                    *[a.ctypes.data_as(_pf) for a in stacked], 1 if feature_offset_quirk else 0,
                                    None if hd is None else hd.ctypes.data_as(_pf), nthreads)
    if rc:
        raise RuntimeError(f"oracle GAT rc={rc}")
    return (out, hd) if dump_h else out


_Q_MODEL_IDS = {"GCN": 2, "GAT": 3, "PNA": 4, "DGN": 5}


def q_forward(model, batch, weight_sets, reload_weights=None, nthreads=1, feature_offset_quirk=False):
    """orc_q_compute_graphs: GCN / GAT / PNA in ap_fixed<16,6>, DGN in ap_fixed<16,3> (q_oracle.c).
    Returns (logits as float, 16-bit patterns)."""
    lib = load()
    model = model.upper()
    G = batch.num_graphs
    if reload_weights is None:
        reload_weights = np.zeros(G, np.int32)
        if G:
            reload_weights[0] = 1
    keys = list(weight_sets[0].keys())
    stacked = [_f(np.stack([np.asarray(ws[k], np.float32) for ws in weight_sets])) for k in keys]
    elems = (C.c_long * len(stacked))(*[int(a[0].size) for a in stacked])
    tens = (_pf * len(stacked))(*[a.ctypes.data_as(_pf) for a in stacked])
    out, out_q = np.zeros(G, np.float32), np.zeros(G, np.int16)
    nn, ne, rw = _i(batch.nums_of_nodes), _i(batch.nums_of_edges), _i(reload_weights)
    nf, el, ea = _i(batch.node_feature), _i(batch.edge_list), _i(batch.edge_attr)
    eig = _f(batch.node_eigen) if batch.node_eigen is not None else None
    lib.orc_q_compute_graphs.argtypes = [C.c_int, C.c_int, _pi, _pi, _pi, _pf, C.POINTER(C.c_int16), _pi, _pf, _pi, _pi, C.c_int,
                                         C.POINTER(_pf), C.POINTER(C.c_long), C.c_int, C.c_int]
    lib.orc_q_compute_graphs.restype = C.c_int
    rc = lib.orc_q_compute_graphs(_Q_MODEL_IDS[model], G, nn.ctypes.data_as(_pi), ne.ctypes.data_as(_pi), rw.ctypes.data_as(_pi),
                                  out.ctypes.data_as(_pf), out_q.ctypes.data_as(C.POINTER(C.c_int16)), nf.ctypes.data_as(_pi),
                                  None if eig is None else eig.ctypes.data_as(_pf), el.ctypes.data_as(_pi), ea.ctypes.data_as(_pi),
                                  len(stacked), tens, elems, 1 if feature_offset_quirk else 0, nthreads)
    if rc:
        raise RuntimeError(f"oracle {model}-Q rc={rc}")
    return out, out_q

"""Independent float64 NumPy restatement of the model equations on the BATCHED super-graph
(globalised ids, scatter-add by destination, segmented mean).  Test infrastructure: it shares
no code with oracle/*.c and no loop structure with the reference, so agreement between the two
checks edge direction, ReLU placement, eps = 0, pooling and the weight layouts.

Equations: SURVEY.md section 3.3 (GIN)."""
import numpy as np

ND_OFF = np.array([0, 119, 123, 135, 147, 157, 163, 169, 171])
ED_OFF = np.array([0, 5, 11])


def gin_forward(batch, w, return_h=False):
    f64 = lambda a: np.asarray(a, dtype=np.float64)
    nemb, eemb = f64(w["node_embedding_weight"]), f64(w["edge_embedding_weight"])
    w1, b1 = f64(w["node_mlp_1_weights"]), f64(w["node_mlp_1_bias"])
    w2, b2 = f64(w["node_mlp_2_weights"]), f64(w["node_mlp_2_bias"])
    pw, pb = f64(w["graph_pred_weights"]).reshape(-1), float(np.asarray(w["graph_pred_bias"]).reshape(-1)[0])
    N = batch.total_nodes
    ge = batch.global_edges()
    u, v = ge[:, 0], ge[:, 1]
    h = nemb[batch.node_feature.astype(np.int64) + ND_OFF[None, :]].sum(axis=1)
    hs = [h]
    for l in range(5):
        ee = eemb[l][batch.edge_attr.astype(np.int64) + ED_OFF[None, :]].sum(axis=1)
        msg = np.maximum(h[u] + ee, 0.0)
        m = np.zeros((N, 100))
        np.add.at(m, v, msg)
        a = m + h
        hid = np.maximum(a @ w1[l].T + b1[l], 0.0)
        h = hid @ w2[l].T + b2[l]
        if l != 4:
            h = np.maximum(h, 0.0)
        hs.append(h)
    off = batch.node_offsets()
    pooled = np.add.reduceat(h, off[:-1], axis=0) / batch.nums_of_nodes[:, None]
    out = pooled @ pw + pb
    return (out, np.stack(hs)) if return_h else out


def gcn_forward(batch, w, return_x=False):
    """GCN equations on the batched super-graph, float64."""
    f64 = lambda a: np.asarray(a, dtype=np.float64)
    nemb, eemb = f64(w["node_embedding_weight"]), f64(w["edge_embedding_weight"])
    cw, cb, root = f64(w["convs_weight"]), f64(w["convs_bias"]), f64(w["convs_root_emb_weight"])
    bnw, bnb, bnm, bnv = f64(w["bn_weight"]), f64(w["bn_bias"]), f64(w["bn_mean"]), f64(w["bn_var"])
    pw, pb = f64(w["graph_pred_weights"]).reshape(-1), float(np.asarray(w["graph_pred_bias"]).reshape(-1)[0])
    N = batch.total_nodes
    ge = batch.global_edges()
    u, v = ge[:, 0], ge[:, 1]
    outdeg = np.bincount(u, minlength=N).astype(np.float64)
    dinv = np.where(outdeg > 0, 1.0 / np.sqrt(outdeg + 1.0), 0.0)
    norm = dinv[u] * dinv[v]
    bn = lambda t, l: (t - bnm[l]) / np.sqrt(bnv[l] + 2.0 ** -10) * bnw[l] + bnb[l]
    h0 = nemb[batch.node_feature.astype(np.int64) + ND_OFF[None, :]].sum(axis=1)
    a = h0
    xs = []
    for l in range(5):
        x = a @ cw[l].T + cb[l]
        xs.append(x)
        ee = eemb[l][batch.edge_attr.astype(np.int64) + ED_OFF[None, :]].sum(axis=1)
        m = np.zeros((N, 100))
        np.add.at(m, v, norm[:, None] * np.maximum(x[u] + ee, 0.0))
        pre = bn(m + np.maximum(x + root[l], 0.0) / (outdeg[:, None] + 1.0), l)
        a = np.maximum(pre, 0.0)
    off = batch.node_offsets()
    pooled = np.add.reduceat(pre, off[:-1], axis=0) / batch.nums_of_nodes[:, None]
    out = pooled @ pw + pb
    return (out, np.stack(xs)) if return_x else out

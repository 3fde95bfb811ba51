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
    pw, pb = f64(w["graph_pred_weights"]).reshape(-1, 100), f64(w["graph_pred_bias"]).reshape(-1)  # [NUM_TASK][100], [NUM_TASK]
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
    out = pooled @ pw.T + pb
    if out.shape[1] == 1:
        out = out[:, 0]
    return (out, np.stack(hs)) if return_h else out


def gcn_forward(batch, w, return_x=False):
    """GCN equations on the batched super-graph, float64."""
    f64 = lambda a: np.asarray(a, dtype=np.float64)
    nemb, eemb = f64(w["node_embedding_weight"]), f64(w["edge_embedding_weight"])
    cw, cb, root = f64(w["convs_weight"]), f64(w["convs_bias"]), f64(w["convs_root_emb_weight"])
    bnw, bnb, bnm, bnv = f64(w["bn_weight"]), f64(w["bn_bias"]), f64(w["bn_mean"]), f64(w["bn_var"])
    pw, pb = f64(w["graph_pred_weights"]).reshape(-1, 100), f64(w["graph_pred_bias"]).reshape(-1)
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
    out = pooled @ pw.T + pb
    if out.shape[1] == 1:
        out = out[:, 0]
    return (out, np.stack(xs)) if return_x else out


def pna_forward(batch, w, return_h=False):
    """PNA equations on the batched super-graph, float64."""
    f64 = lambda a: np.asarray(a, dtype=np.float64)
    nemb, cw, cb = f64(w["node_embedding_weight"]), f64(w["node_conv_weights"]), f64(w["node_conv_bias"])
    avg = float(np.asarray(w["avg_deg"]).reshape(-1)[0])
    N = batch.total_nodes
    ge = batch.global_edges()
    u, v = ge[:, 0], ge[:, 1]
    indeg = np.bincount(v, minlength=N).astype(np.float64)
    outdeg = np.bincount(u, minlength=N).astype(np.float64)
    logd = np.log(outdeg + 1.0)
    t = logd / avg
    scale = np.where(logd == 0, 1.0, avg / np.where(logd == 0, 1.0, logd))
    sf = np.stack([np.ones(N), t, scale], axis=1)  # [N, 3]
    deg1 = np.maximum(indeg, 1.0)[:, None]
    h = nemb[batch.node_feature.astype(np.int64) + ND_OFF[None, :]].sum(axis=1)
    hs = [h]
    for l in range(4):
        x = h[u]
        S = np.zeros((N, 80)); Q = np.zeros((N, 80))
        np.add.at(S, v, x); np.add.at(Q, v, x * x)
        mn = np.full((N, 80), 31.9990234375); mx = np.full((N, 80), -32.0)
        np.minimum.at(mn, v, x); np.maximum.at(mx, v, x)
        mean = S / deg1
        std = np.sqrt(np.maximum(Q / deg1 - mean * mean, 0.0))
        agg = np.stack([mean, mn, mx, std], axis=1)  # [N, 4(aggr enum order), 80]
        y = np.einsum("osai,nai->nso", cw[l], agg)    # [N, 3, 80]
        acc = cb[l] + (y * sf[:, :, None]).sum(axis=1)
        h = h + np.maximum(acc, 0.0)
        hs.append(h)
    off = batch.node_offsets()
    hg = np.add.reduceat(h, off[:-1], axis=0) / batch.nums_of_nodes[:, None]
    o1 = np.maximum(hg @ f64(w["graph_mlp_1_weights"]).T + f64(w["graph_mlp_1_bias"]), 0.0)
    o2 = np.maximum(o1 @ f64(w["graph_mlp_2_weights"]).T + f64(w["graph_mlp_2_bias"]), 0.0)
    out = o2 @ f64(w["graph_mlp_3_weights"]).reshape(-1) + float(np.asarray(w["graph_mlp_3_bias"]).reshape(-1)[0])
    return (out, np.stack(hs)) if return_h else out


def dgn_forward(batch, w, return_h=False):
    """DGN equations on the batched super-graph, float64; x / 0 = 0 for the out-degree divide."""
    f64 = lambda a: np.asarray(a, dtype=np.float64)
    emb = f64(w["embedding_h_atom_embedding_list_weights"])
    lw, lb = f64(w["layers_posttrans_fully_connected_0_linear_weight"]), f64(w["layers_posttrans_fully_connected_0_linear_bias"])
    N = batch.total_nodes
    ge = batch.global_edges()
    u, v = ge[:, 0], ge[:, 1]
    eig1 = f64(batch.node_eigen)[:, 1]
    we = eig1[u] - eig1[v]
    wsum = np.bincount(v, weights=we, minlength=N)
    abssum = np.bincount(v, weights=np.abs(we), minlength=N)
    abssum = np.where(abssum == 0, 2.0 ** -13, abssum)
    outdeg = np.bincount(u, minlength=N).astype(np.float64)
    h = emb[np.arange(9)[None, :], batch.node_feature.astype(np.int64)].sum(axis=1)
    hs = [h]
    for l in range(4):
        m1 = np.zeros((N, 100)); m2 = np.zeros((N, 100))
        np.add.at(m1, v, h[u]); np.add.at(m2, v, h[u] * we[:, None])
        a1 = np.where(outdeg[:, None] == 0, 0.0, m1 / np.maximum(outdeg, 1.0)[:, None])
        a2 = np.abs((m2 - wsum[:, None] * h) / abssum[:, None])
        W = lw[l].reshape(100, 2, 100)
        acc = lb[l] + a1 @ W[:, 0, :].T + a2 @ W[:, 1, :].T
        h = h + np.maximum(acc, 0.0)
        hs.append(h)
    off = batch.node_offsets()
    hg = np.add.reduceat(h, off[:-1], axis=0) / batch.nums_of_nodes[:, None]
    o1 = np.maximum(hg @ f64(w["MLP_layer_FC_layers_0_weight"]).T + f64(w["MLP_layer_FC_layers_0_bias"]), 0.0)
    o2 = np.maximum(o1 @ f64(w["MLP_layer_FC_layers_1_weight"]).T + f64(w["MLP_layer_FC_layers_1_bias"]), 0.0)
    out = o2 @ f64(w["MLP_layer_FC_layers_2_weight"]).reshape(-1) + float(np.asarray(w["MLP_layer_FC_layers_2_bias"]).reshape(-1)[0])
    return (out, np.stack(hs)) if return_h else out


def gat_forward(batch, w, return_h=False):
    """GAT equations on the batched super-graph, float64, per-graph feature offsets applied.
    Feature index f = dim * 4 + head."""
    f64 = lambda a: np.asarray(a, dtype=np.float64)
    tgt, srcw = f64(w["scoring_fn_target"]), f64(w["scoring_fn_source"])      # [5][4 head][16 dim]
    lin, skip = f64(w["linear_proj_weights"]), f64(w["skip_proj_weights"])    # [5][ho][do][hi][di]
    N = batch.total_nodes
    ge = batch.global_edges()
    u = np.concatenate([np.arange(N), ge[:, 0]])   # self edge for every node
    v = np.concatenate([np.arange(N), ge[:, 1]])
    mat = lambda t, l: t[l].transpose(1, 0, 3, 2).reshape(64, 64)   # [do, ho, di, hi] -> rows do*4+ho, cols di*4+hi
    feat = batch.node_feature.astype(np.float64)
    skipin = np.zeros((N, 16, 4)); skipin[:, :9, 0] = feat
    skipin = skipin.reshape(N, 64)
    proj = skipin @ mat(lin, 0).T
    outs = []
    for l in range(5):
        p3 = proj.reshape(N, 16, 4)
        ssrc = np.einsum("ndh,hd->nh", p3, srcw[l]); stgt = np.einsum("ndh,hd->nh", p3, tgt[l])
        s = ssrc[v] + stgt[u]
        e = np.exp(np.where(s < 0, 0.2 * s, s))
        den = np.zeros((N, 4)); np.add.at(den, v, e)
        num = np.zeros((N, 16, 4)); np.add.at(num, v, e[:, None, :] * p3[u])
        msg = (num / den[:, None, :]).reshape(N, 64)
        o = msg + skipin @ mat(skip, l).T
        if l == 4:
            emb = o.reshape(N, 16, 4).mean(axis=2)
            break
        o = np.where(o <= 0, np.exp(o) - 1.0, o)
        outs.append(o)
        skipin = o
        proj = o @ mat(lin, l + 1).T
    off = batch.node_offsets()
    hg = np.add.reduceat(emb, off[:-1], axis=0) / batch.nums_of_nodes[:, None]
    out = hg @ f64(w["graph_pred_weights"]).reshape(-1) + float(np.asarray(w["graph_pred_bias"]).reshape(-1)[0])
    return (out, np.stack(outs)) if return_h else out


# --------------------------------------------------------------------------- GIN in ap_fixed<16,6> (Q6.10)
def _q(x):
    """float -> 16-bit pattern: floor(x * 1024), low 16 bits, sign extended (as int64)."""
    i = np.floor(np.asarray(x, np.float64) * 1024.0).astype(np.int64)
    return _wrap16(i)


def _wrap16(i):
    return ((np.asarray(i, np.int64) + 32768) & 0xFFFF) - 32768


def _dense_floor(a, wq, bq):
    """out[n][o] = wrap16(b[o] + sum_k floor(a[n][k] * w[o][k] / 1024)): every product truncated on its own."""
    out = np.empty((a.shape[0], wq.shape[0]), np.int64)
    for o in range(wq.shape[0]):  # row by row keeps the [n][k] temporary small
        out[:, o] = ((a * wq[o][None, :]) >> 10).sum(axis=1)
    return _wrap16(out + bq[None, :])


def gin_forward_q(batch, w, return_h=False):
    """Independent vectorised restatement of oracle/ginq_oracle.c on the batched super-graph (int64 NumPy; arithmetic
    mod 2^16, so the order of the sums does not matter).  Returns the 16-bit logit patterns."""
    nemb, eemb = _q(w["node_embedding_weight"]), _q(w["edge_embedding_weight"])
    w1, b1, w2, b2 = _q(w["node_mlp_1_weights"]), _q(w["node_mlp_1_bias"]), _q(w["node_mlp_2_weights"]), _q(w["node_mlp_2_bias"])
    pw, pb = _q(w["graph_pred_weights"]).reshape(-1), int(_q(w["graph_pred_bias"]).reshape(-1)[0])
    N = batch.total_nodes
    ge = batch.global_edges()
    u, v = ge[:, 0], ge[:, 1]
    h = _wrap16(nemb[batch.node_feature.astype(np.int64) + ND_OFF[None, :]].sum(axis=1))
    hs = [h]
    for l in range(5):
        ee = _wrap16(eemb[l][batch.edge_attr.astype(np.int64) + ED_OFF[None, :]].sum(axis=1))
        msg = np.maximum(_wrap16(h[u] + ee), 0)
        m = np.zeros((N, 100), np.int64)
        np.add.at(m, v, msg)
        a = _wrap16(m + h)
        hid = np.maximum(_dense_floor(a, w1[l], b1[l]), 0)
        h = _dense_floor(hid, w2[l], b2[l])
        if l != 4:
            h = np.maximum(h, 0)
        hs.append(h)
    off = batch.node_offsets()
    sums = _wrap16(np.add.reduceat(h, off[:-1], axis=0))
    hg = _wrap16(np.floor_divide(sums, np.asarray(batch.nums_of_nodes, np.int64)[:, None]))
    out = _wrap16(((hg * pw[None, :]) >> 10).sum(axis=1) + pb)
    return (out, np.stack(hs)) if return_h else out

"""Exporter: OGB-style state_dicts and PyG-style graphs -> the reference's file formats.
CPU only.  The "OGB side" below is an independent float64 restatement of the OGB example models' inference
semantics (ogb/examples/graphproppred/mol/conv.py: GINConv / GCNConv with BatchNorm in eval mode), so the tests
check the parameter-name mapping, the table concatenation, the BatchNorm folding (GIN) and the variance shift
(GCN) through tests/numpy_ref.py, which restates the REFERENCE's equations."""
import os

import numpy as np
import pytest

from flowgnn_amd import export, graphpack as gp, weights
from tests import numpy_ref

L, D = 5, 100


def _bn_params(rng, prefix, n, sd):
    sd[prefix + ".weight"] = 1.0 + 0.2 * rng.standard_normal(n)
    sd[prefix + ".bias"] = 0.2 * rng.standard_normal(n)
    sd[prefix + ".running_mean"] = 0.3 * rng.standard_normal(n)
    sd[prefix + ".running_var"] = rng.uniform(0.3, 1.5, n)
    sd[prefix + ".num_batches_tracked"] = np.array(7)


def ogb_state_dict(kind, seed, with_bn=True, tasks=1):
    rng = np.random.default_rng(seed)
    sd = {}
    for k, n in enumerate(export.ATOM_DIMS):
        sd[f"gnn_node.atom_encoder.atom_embedding_list.{k}.weight"] = 0.11 * rng.standard_normal((n, D))
    for l in range(L):
        c = f"gnn_node.convs.{l}"
        for k, n in enumerate(export.BOND_DIMS):
            sd[f"{c}.bond_encoder.bond_embedding_list.{k}.weight"] = 0.14 * rng.standard_normal((n, D))
        if kind == "gin":
            sd[f"{c}.eps"] = np.zeros(1)
            sd[f"{c}.mlp.0.weight"] = 0.075 * rng.standard_normal((2 * D, D))
            sd[f"{c}.mlp.0.bias"] = 0.5 * rng.standard_normal(2 * D)
            last = 3 if with_bn else 2
            if with_bn:
                _bn_params(rng, f"{c}.mlp.1", 2 * D, sd)
            sd[f"{c}.mlp.{last}.weight"] = 0.053 * rng.standard_normal((D, 2 * D))
            sd[f"{c}.mlp.{last}.bias"] = 0.3 * rng.standard_normal(D)
        else:
            sd[f"{c}.linear.weight"] = 0.09 * rng.standard_normal((D, D))
            sd[f"{c}.linear.bias"] = 0.1 * rng.standard_normal(D)
            sd[f"{c}.root_emb.weight"] = 0.3 * rng.standard_normal((1, D))
        if with_bn or kind == "gcn":
            _bn_params(rng, f"gnn_node.batch_norms.{l}", D, sd)
    sd["graph_pred_linear.weight"] = 0.15 * rng.standard_normal((tasks, D))
    sd["graph_pred_linear.bias"] = 0.1 * rng.standard_normal(tasks)
    return sd


def _bn(sd, prefix, t):
    return (t - sd[prefix + ".running_mean"]) / np.sqrt(sd[prefix + ".running_var"] + 1e-5) * sd[prefix + ".weight"] + sd[prefix + ".bias"]


def ogb_forward(kind, sd, batch):
    """Inference of the OGB example GNN (JK='last', no residual, mean pooling), float64."""
    ge = batch.global_edges()
    row, col = ge[:, 0], ge[:, 1]  # messages flow row -> col
    N = batch.total_nodes
    h = sum(sd[f"gnn_node.atom_encoder.atom_embedding_list.{k}.weight"][batch.node_feature[:, k]] for k in range(9))
    for l in range(L):
        c = f"gnn_node.convs.{l}"
        e = sum(sd[f"{c}.bond_encoder.bond_embedding_list.{k}.weight"][batch.edge_attr[:, k]] for k in range(3))
        if kind == "gin":
            agg = np.zeros((N, D))
            np.add.at(agg, col, np.maximum(h[row] + e, 0.0))
            z = ((1.0 + sd[f"{c}.eps"][0]) * h + agg) @ sd[f"{c}.mlp.0.weight"].T + sd[f"{c}.mlp.0.bias"]
            if f"{c}.mlp.1.weight" in sd:
                z = np.maximum(_bn(sd, f"{c}.mlp.1", z), 0.0) @ sd[f"{c}.mlp.3.weight"].T + sd[f"{c}.mlp.3.bias"]
            else:
                z = np.maximum(z, 0.0) @ sd[f"{c}.mlp.2.weight"].T + sd[f"{c}.mlp.2.bias"]
        else:
            x = h @ sd[f"{c}.linear.weight"].T + sd[f"{c}.linear.bias"]
            deg = np.bincount(row, minlength=N) + 1.0
            norm = deg[row] ** -0.5 * deg[col] ** -0.5
            z = np.zeros((N, D))
            np.add.at(z, col, norm[:, None] * np.maximum(x[row] + e, 0.0))
            z = z + np.maximum(x + sd[f"{c}.root_emb.weight"], 0.0) / deg[:, None]
        if f"gnn_node.batch_norms.{l}.weight" in sd:
            z = _bn(sd, f"gnn_node.batch_norms.{l}", z)
        h = z if l == L - 1 else np.maximum(z, 0.0)
    off = batch.node_offsets()
    pooled = np.add.reduceat(h, off[:-1], axis=0) / batch.nums_of_nodes[:, None]
    out = pooled @ sd["graph_pred_linear.weight"].T + sd["graph_pred_linear.bias"]
    return out[:, 0] if out.shape[1] == 1 else out


@pytest.mark.parametrize("with_bn", [True, False])
def test_gin_export_matches_ogb_semantics(tmp_path, with_bn):
    sd = ogb_state_dict("gin", 3, with_bn=with_bn)
    batch = gp.synth_molhiv_batch(24, seed=9)
    want = ogb_forward("gin", sd, batch)
    export.export_weights("GIN", sd, str(tmp_path))
    assert sorted(os.listdir(tmp_path)) == sorted([f for f, _ in weights.GIN_FILES.values()] + ["gin_ep1_eps_dim100.bin"])
    w = weights.load_gin_weights(str(tmp_path))  # what host / flowgnn_load_weights_dir would read
    got = numpy_ref.gin_forward(batch, w)
    assert np.allclose(got, want, rtol=1e-4, atol=2e-5), np.abs(got - want).max()  # float32 files vs float64 source


def test_gcn_export_matches_ogb_semantics(tmp_path):
    sd = ogb_state_dict("gcn", 4)
    batch = gp.synth_molpcba_batch(24, seed=10)
    want = ogb_forward("gcn", sd, batch)
    export.export_weights("GCN", sd, str(tmp_path))
    assert os.path.getsize(tmp_path / weights.GCN_FILE) == 76906 * 4
    got = numpy_ref.gcn_forward(batch, weights.load_gcn_weights(str(tmp_path)))
    assert np.allclose(got, want, rtol=1e-4, atol=2e-5), np.abs(got - want).max()


def test_gcn_file_is_the_flattened_state_dict(tmp_path):
    """GCN/src/host_load.cc's offsets are those of the state_dict tensors laid end to end (401 floats per BatchNorm)."""
    sd = ogb_state_dict("gcn", 5)
    export.export_weights("GCN", sd, str(tmp_path))
    flat = np.fromfile(tmp_path / weights.GCN_FILE, dtype="<f4")
    assert np.allclose(flat[:119 * D], sd["gnn_node.atom_encoder.atom_embedding_list.0.weight"].ravel().astype(np.float32))
    assert np.allclose(flat[17300 + 11500 * 2:17300 + 11500 * 2 + D * D], sd["gnn_node.convs.2.linear.weight"].ravel().astype(np.float32))
    assert np.allclose(flat[17300 + 11500 * 3 + 10100:17300 + 11500 * 3 + 10200], sd["gnn_node.convs.3.root_emb.weight"].ravel().astype(np.float32))
    assert np.allclose(flat[74800 + 401 * 4 + 200:74800 + 401 * 4 + 300], sd["gnn_node.batch_norms.4.running_mean"].astype(np.float32))
    assert np.allclose(flat[76805:76905], sd["graph_pred_linear.weight"].ravel().astype(np.float32))


@pytest.mark.parametrize("kind", ["gin", "gcn"])
def test_multi_task_head_molpcba(tmp_path, kind, oracle):
    """ogbg-molpcba has 128 tasks: the exporter keeps the whole head ([128][100] + [128]), the files round-trip, and the C
    oracle's NUM_TASK-dimensioned readout (GIN/src/dcl.h:25,80; linear.cc:26-47) agrees with the OGB model's semantics."""
    sd = ogb_state_dict(kind, 11, tasks=128)
    batch = gp.synth_molpcba_batch(20, seed=12)
    want = ogb_forward(kind, sd, batch)
    assert want.shape == (20, 128)
    export.export_weights(kind.upper(), sd, str(tmp_path), multi_task=True)
    if kind == "gin":
        w = weights.load_gin_weights(str(tmp_path), num_tasks=128)
        got_np, got_c = numpy_ref.gin_forward(batch, w), oracle.gin_forward(batch, [w], num_tasks=128)
    else:
        assert os.path.getsize(tmp_path / weights.GCN_FILE) == (76805 + 101 * 128) * 4
        w = weights.load_gcn_weights(str(tmp_path), num_tasks=128)
        got_np, got_c = numpy_ref.gcn_forward(batch, w), oracle.gcn_forward(batch, [w], num_tasks=128)
    assert w["graph_pred_weights"].shape == (128, 100) and w["graph_pred_bias"].shape == (128,)
    assert np.allclose(got_np, want, rtol=1e-4, atol=2e-5), np.abs(got_np - want).max()
    assert got_c.shape == (20, 128) and np.allclose(got_c, want, rtol=1e-4, atol=1e-4), np.abs(got_c - want).max()


def test_torch_tensors_are_accepted(tmp_path):
    torch = pytest.importorskip("torch")
    sd = {k: torch.tensor(v, requires_grad=np.issubdtype(np.asarray(v).dtype, np.floating) and "running" not in k and "num_b" not in k)
          for k, v in ogb_state_dict("gin", 6).items()}
    a = export.gin_weights_from_ogb_state_dict(sd)
    b = export.gin_weights_from_ogb_state_dict(ogb_state_dict("gin", 6))
    assert all(np.array_equal(a[k], b[k]) for k in a)


def test_refusals():
    sd = ogb_state_dict("gin", 7)
    sd["gnn_node.convs.2.eps"] = np.array([0.05])
    with pytest.raises(export.ExportError, match="eps"):
        export.gin_weights_from_ogb_state_dict(sd)
    export.gin_weights_from_ogb_state_dict(sd, eps_tol=0.1)  # the caller may accept the approximation
    with pytest.raises(export.ExportError, match="NUM_TASK"):
        export.gin_weights_from_ogb_state_dict(ogb_state_dict("gin", 7, tasks=128))
    bad = ogb_state_dict("gcn", 7)
    del bad["gnn_node.convs.1.root_emb.weight"]
    with pytest.raises(export.ExportError, match="root_emb"):
        export.gcn_weights_from_ogb_state_dict(bad)
    with pytest.raises(export.ExportError, match="PNA"):
        export.export_weights("PNA", {}, "/tmp/unused")


class _Data:  # the attributes of a torch_geometric.data.Data that matter
    def __init__(self, x, edge_index, edge_attr, eig=None):
        self.x, self.edge_index, self.edge_attr, self.eig = x, edge_index, edge_attr, eig


def test_dataset_round_trip(tmp_path):
    src = gp.synth_hep10k_batch(5, seed=11, with_eigen=True)
    no, eo = src.node_offsets(), src.edge_offsets()
    graphs = [_Data(src.node_feature[no[g]:no[g + 1]], src.edge_list[eo[g]:eo[g + 1]].T, src.edge_attr[eo[g]:eo[g + 1]],
                    src.node_eigen[no[g]:no[g + 1]]) for g in range(src.num_graphs)]
    graphs[2] = {"x": graphs[2].x, "edge_index": graphs[2].edge_index, "edge_attr": graphs[2].edge_attr, "eig": graphs[2].eig}
    made = export.export_dataset(graphs, str(tmp_path / "graphs"), eig_dir=str(tmp_path / "eig"))
    back = gp.read_pack(str(tmp_path / "graphs"), eig_dir=str(tmp_path / "eig"))
    for a, b in ((made, src), (back, src)):
        assert np.array_equal(a.nums_of_nodes, b.nums_of_nodes) and np.array_equal(a.nums_of_edges, b.nums_of_edges)
        assert np.array_equal(a.node_feature, b.node_feature) and np.array_equal(a.edge_list, b.edge_list)
        assert np.array_equal(a.edge_attr, b.edge_attr)
    assert np.allclose(back.node_eigen, src.node_eigen, rtol=1e-4, atol=1e-7)  # the text format keeps 5 digits
    with pytest.raises(export.ExportError, match="9 integer"):
        export.batch_from_graphs([_Data(np.zeros((3, 8), int), np.zeros((2, 0), int), None)])
    with pytest.raises(export.ExportError, match="outside"):
        export.batch_from_graphs([_Data(np.zeros((3, 9), int), np.array([[0], [3]]), None)])

"""Re-create batch (seed, iteration) of scripts/dev/fuzz.py and show where the HIP path and the oracle differ.  usage: fuzz_case.py MODEL seed iter"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
model, seed0, it = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
src = open(os.path.join(ROOT, "scripts/dev/fuzz.py")).read()
pre = src[:src.index("e = Engine(model, 0)")]
pre = pre.replace('model = sys.argv[1] if len(sys.argv) > 1 else "GIN"', f'model = "{model}"').replace("seed0 = int(sys.argv[3]) if len(sys.argv) > 3 else 1", f"seed0 = {seed0}")
pre = pre.replace("budget = float(sys.argv[2]) if len(sys.argv) > 2 else 60.0", "budget = 0")
g = {"__file__": os.path.join(ROOT, "scripts/dev/fuzz.py")}
exec(pre, g)
gp, rand_graph, w, ofn, Engine = g["gp"], g["rand_graph"], g["w"], g["ofn"], g["Engine"]
rng = np.random.default_rng(seed0 * 100003 + it)
graphs = [rand_graph(rng) for _ in range(int(rng.integers(1, 40)))]
if rng.random() < 0.3:
    mol = gp.synth_molhiv_batch(int(rng.integers(1, 300)), seed=int(rng.integers(1 << 30)))
    if model == "DGN":
        eg = np.zeros((mol.total_nodes, 4), np.float32); eg[:, 1] = rng.uniform(-1, 1, mol.total_nodes)
        mol = gp.GraphBatch(mol.nums_of_nodes, mol.nums_of_edges, mol.node_feature, mol.edge_list, mol.edge_attr, eg)
    graphs.insert(int(rng.integers(0, len(graphs) + 1)), mol)
b = gp.concat_batches(graphs)
if model == "GIN-VN":
    b = gp.add_virtual_nodes(b)
want, hd = ofn(b, [w], dump_h=True, nthreads=8)
print("graphs", b.num_graphs, "sizes", b.nums_of_nodes, "edges", b.nums_of_edges, "scale", float(np.abs(hd).max()))
for opts in ({}, {"dgn_mfma_agg": 0}):
    e = Engine(model, 0, options=opts); e.set_weights(w)
    got = e.forward(b); h = e.final_h()
    d = np.abs(got - want)
    dh = np.abs(h - hd[-1]).max(axis=1)
    print(opts, "logits max err", d.max(), "at graph", int(d.argmax()), "| rows max err", dh.max(), "at node", int(dh.argmax()))
    if model == "DGN" and not opts:
        v = int(dh.argmax()); off = np.concatenate([[0], np.cumsum(b.nums_of_nodes)]); gi = int(np.searchsorted(off, v, side="right") - 1)
        ge = b.global_edges(); ins = ge[ge[:, 1] == v][:, 0]
        eig = b.node_eigen[:, 1]
        print("  node", v, "of graph", gi, "in-edges from", np.sort(ins)[:20], "eig_v", eig[v], "weights", (eig[ins] - eig[v])[:20], "sum|w|", np.abs(eig[ins] - eig[v]).sum())
    e.close()
if model == "PNA":
    for opts in ({"pna_fused": 0}, {"pna_mfma": 32}, {"pna_fused": 0, "pna_mfma": 32}):
        e = Engine(model, 0, options=opts); e.set_weights(w)
        got = e.forward(b); h = e.final_h()
        print(opts, "logits max err", np.abs(got - want).max(), "rows max err", np.abs(h - hd[-1]).max(), "max|h|", np.abs(h).max(), "exact_reruns", e.exact_reruns())
        e.close()
    print("oracle max|h| per layer", [float(np.abs(x).max()) for x in hd])

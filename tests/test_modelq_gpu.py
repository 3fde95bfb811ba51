"""Bit-faithful fixed-point modes of GCN / GAT / PNA (ap_fixed<16,6>) and DGN (ap_fixed<16,3>): the HIP path (modelq.hip)
must reproduce oracle/q_oracle.c bit for bit -- every output is a 16-bit pattern, compared with ==, on molecule-, kNN- and
degenerate-shaped batches, under batch permutation, and through the C++ host binary."""
import os
import subprocess

import numpy as np
import pytest

from flowgnn_amd import Engine, graphpack as gp, weights
from tests.test_oracle_dgn import with_eigen
from tests.test_oracle_gcn import directed_variant

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FRAC = {"GCN": 10, "GAT": 10, "PNA": 10, "DGN": 13}


def batches(model):
    eig = model == "DGN"
    out = [gp.synth_molhiv_batch(150, seed=3), gp.synth_molpcba_batch(100, seed=4), gp.synth_hep10k_batch(24, seed=5, with_eigen=False),
           directed_variant(gp.synth_molhiv_batch(40, seed=12))]
    # degenerate graphs: single node without edges, two nodes one edge, a 33-node graph without edges
    nn, ne = np.array([1, 2, 33], np.int32), np.array([0, 1, 0], np.int32)
    nf = np.zeros((36, 9), np.int32)
    nf[:, 0] = np.arange(36) % 119
    out.append(gp.GraphBatch(nn, ne, nf, np.array([[1, 0]], np.int32), np.array([[4, 5, 1]], np.int32)))
    return [with_eigen(b, 1) if eig else b for b in out]


def q_engine(model, w):
    e = Engine(model, device=0)
    e.set_weights(w)
    e.set_numeric_mode("q6.10")
    return e


@pytest.mark.parametrize("model", ["GCN", "GAT", "PNA", "DGN"])
def test_bit_exact_vs_q_oracle(oracle, model):
    w = weights.SYNTH[model](seed=7)
    e = q_engine(model, w)
    scale = float(1 << FRAC[model])
    for b in batches(model):
        got = e.forward(b)
        want, want_q = oracle.q_forward(model, b, [w], nthreads=8)
        assert np.array_equal(np.round(got * scale).astype(np.int64), want_q.astype(np.int64)), \
            (model, np.abs(got - want).max(), int((got != want).sum()), b.num_graphs)
        assert np.array_equal(got, want)
    # graphs are independent and all sums are order-free: any batch order gives the same patterns
    b = batches(model)[0]
    out = e.forward(b)
    perm = np.random.default_rng(0).permutation(b.num_graphs)
    shuffled = gp.concat_batches([b.slice(int(g), int(g) + 1) for g in perm])
    assert np.array_equal(e.forward(shuffled), out[perm])
    # back to fp32 on the same engine
    e.set_numeric_mode("f32")
    f = e.forward(b)
    ref = getattr(oracle, model.lower() + "_forward")(b, [w], nthreads=8)
    assert np.allclose(f, ref, rtol=2e-4, atol=2e-3 * max(1.0, float(np.abs(ref).max())))
    e.close()


def test_second_weight_set_and_trained_scale_weights(oracle):
    """Weights with larger magnitudes (sums that really wrap) and a weight reload on a live engine."""
    for model in ("GCN", "PNA", "DGN", "GAT"):
        w = weights.SYNTH[model](seed=11)
        big = {k: (np.asarray(v) * np.float32(3.0) if "weight" in k or "conv" in k else np.asarray(v)) for k, v in w.items()}
        b = batches(model)[2]
        e = q_engine(model, w)
        assert np.array_equal(e.forward(b), oracle.q_forward(model, b, [w], nthreads=8)[0])
        e.set_weights(big)
        assert np.array_equal(e.forward(b), oracle.q_forward(model, b, [big], nthreads=8)[0])
        e.close()


def test_gat_reference_feature_offset_quirk(monkeypatch, oracle):
    monkeypatch.setenv("FLOWGNN_GAT_REFERENCE_QUIRK", "1")
    w = weights.synth_gat_weights(seed=7)
    b = gp.synth_molhiv_batch(30, seed=9)
    e = q_engine("GAT", w)
    assert np.array_equal(e.forward(b), oracle.q_forward("GAT", b, [w], feature_offset_quirk=True)[0])
    e.close()


def test_host_binary_numeric_flag(tmp_path, oracle):
    """`host <MODEL> --numeric q6.10` (the reference's own arithmetic) for a model other than GIN."""
    model = "DGN"
    w = weights.SYNTH[model](seed=7)
    b = gp.synth_hep10k_batch(6, seed=3)
    gp.write_pack(b, str(tmp_path / "graphs"), eig_dir=str(tmp_path / "eig"))
    weights.SAVERS[model](w, str(tmp_path / "w"))
    out = tmp_path / "HLS_output.txt"
    r = subprocess.run([os.path.join(ROOT, "flowgnn_amd", "host"), model, "--graphs", str(tmp_path / "graphs"), "--weights", str(tmp_path / "w"),
                        "--eig", str(tmp_path / "eig"), "--num-graphs", "6", "--trials", "1", "--numeric", "q6.10", "--out", str(out)],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    got = np.array([float(ln.split(":")[1]) for ln in open(out).read().strip().splitlines()], np.float32)
    seen = gp.read_pack(str(tmp_path / "graphs"), 6, eig_dir=str(tmp_path / "eig"))  # the eigenvectors as the text files carry them
    want = oracle.q_forward(model, seen, [weights.load_dgn_weights(str(tmp_path / "w"))])[0]
    assert np.allclose(got, want, atol=1e-7)  # printed with 8 decimals; the patterns are multiples of 2^-13

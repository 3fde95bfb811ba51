"""The TRAINED weight sets the reference ships (tests/golden/ref_weights_<model>.npz, made by tests/golden/make_ref_weights.py from the
reference's .bin files) on the GPU: the synthetic sets have the shipped sets' per-tensor scales, but not their structure -- GIN logits of
-3 .. -6, PNA's saturating readout head, BatchNorm statistics of a trained GCN.  CPU: the fixtures are what the loaders read from the
reference (when it is present) and the oracle reproduces the `logits_reference_weights` of the per-model fixtures from them.  GPU: the
engine does, through the C ABI, on the graph-resident / fused kernels and on the per-layer ones; the fixed-point modes stay bit-exact."""
import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))
import make_modes_golden as mk  # noqa: E402  (fixture_batch)
from tests.parity import assert_close, oracle_scale  # noqa: E402

MODELS = ["GIN", "GCN", "GAT", "PNA", "DGN"]
FRAC = {"GIN": 10, "GCN": 10, "GAT": 10, "PNA": 10, "DGN": 13}
SWITCH = {"GIN": "gin_resident", "GCN": "gcn_resident", "GAT": "gat_resident", "PNA": "pna_fused", "DGN": "dgn_fused"}


def trained(model):
    z = np.load(os.path.join(HERE, "golden", f"ref_weights_{model.lower()}.npz"))
    return {k: z[k] for k in z.files}


def expected(model):
    return np.load(os.path.join(HERE, "golden", mk.FIXTURES[model] + ".npz"))["logits_reference_weights"]


@pytest.mark.parametrize("model", MODELS)
def test_fixture_is_the_shipped_set_and_the_oracle_reproduces_its_logits(model, oracle):
    from flowgnn_amd import weights
    w = trained(model)
    ref = os.path.join(os.environ.get("FLOWGNN_REFERENCE", "/root/reference"), model)
    if os.path.isdir(ref):
        shipped = weights.LOADERS[model](ref)
        assert sorted(shipped) == sorted(w)
        assert all(np.array_equal(np.asarray(shipped[k], np.float32), w[k]) for k in w)
    got = getattr(oracle, model.lower() + "_forward")(mk.fixture_batch(model), [w])
    want = expected(model)
    assert np.allclose(got, want, rtol=1e-6, atol=1e-6), np.abs(got - want).max()
    assert np.isfinite(want).all() and float(np.abs(want).max()) > 0.5  # trained logits, not noise around zero


@pytest.mark.gpu
@pytest.mark.parametrize("per_layer", [False, True], ids=["resident-or-fused", "per-layer"])
@pytest.mark.parametrize("model", MODELS)
def test_gpu_matches_the_oracle_on_trained_weights(model, per_layer):
    from flowgnn_amd import Engine
    e = Engine(model, device=0, options={SWITCH[model]: 0} if per_layer else {})
    try:
        e.set_weights(trained(model))
        got = e.forward(mk.fixture_batch(model))
    finally:
        e.close()
    want = expected(model)
    assert_close(got, want, what=model)


DATASET = {"GIN": ("molhiv", 4113), "GAT": ("molhiv", 4113), "GCN": ("molpcba", 43773), "PNA": ("hep10k", 10000), "DGN": ("hep10k", 10000)}


@pytest.mark.gpu
@pytest.mark.parametrize("model", MODELS)
def test_trained_weights_at_dataset_size(model, oracle):
    """The shipped TRAINED set of every model on a batch of its own dataset's size and shape (BASELINE configs 2-5: 4 113 molhiv,
    43 773 molpcba, 10 000 hep10k graphs), every graph compared with the oracle.  Trained GIN's activations grow on dense graphs, so the
    scale of the comparison is the oracle's own largest activation (from a 256-graph slice: a full dump is up to 1.7 GB)."""
    from flowgnn_amd import Engine, graphpack as gp
    shape, n = DATASET[model]
    kw = {"with_eigen": model == "DGN"} if shape == "hep10k" else {}
    b = getattr(gp, f"synth_{shape}_batch")(n, seed=4321, **kw)
    w = trained(model)
    fwd = getattr(oracle, model.lower() + "_forward")
    want = fwd(b, [w], nthreads=16)
    _, hd = fwd(b.slice(0, 256), [w], dump_h=True, nthreads=16)
    assert np.isfinite(want).all()
    e = Engine(model, device=0)
    try:
        e.set_weights(w)
        got = e.forward(b)
        reruns = e.exact_reruns()
    finally:
        e.close()
    assert got.shape == (n,)
    assert_close(got, want, oracle_scale(hd), what=(model, n, "reruns", reruns))


@pytest.mark.gpu
def test_gin_vn_on_trained_weights(oracle):
    from flowgnn_amd import Engine, graphpack as gp
    w = trained("GIN")  # GIN-VN ships the same files
    b = gp.add_virtual_nodes(mk.fixture_batch("GIN"))
    want = oracle.gin_forward(b, [w], nthreads=8)
    e = Engine("GIN-VN", device=0)
    try:
        e.set_weights(w)
        got = e.forward(b)
    finally:
        e.close()
    assert np.isfinite(want).all()
    _, hd = oracle.gin_forward(b, [w], dump_h=True, nthreads=8)
    assert_close(got, want, oracle_scale(hd), what="GIN-VN")


@pytest.mark.gpu
@pytest.mark.parametrize("model", MODELS)
def test_fixed_point_mode_bit_exact_on_trained_weights(model, oracle):
    from flowgnn_amd import Engine
    w, b = trained(model), mk.fixture_batch(model)
    if model == "GIN":
        _, want_q = oracle.gin_forward_q(b, [w], nthreads=8)
    else:
        _, want_q = oracle.q_forward(model, b, [w], nthreads=8)
    e = Engine(model, device=0)
    try:
        e.set_weights(w)
        e.set_numeric_mode("q6.10")
        got = e.forward(b)
    finally:
        e.close()
    assert np.array_equal(np.round(got * float(1 << FRAC[model])).astype(np.int64), want_q.astype(np.int64))


# ---------------------------------------------------------------- trained weights beyond the models' own dataset shapes
def _big_graph(n, m, seed):
    """One connected graph at the reference's caps (GIN/src/dcl.h:17-18: 500 nodes / 5 500 edges)."""
    from flowgnn_amd import graphpack as gp
    rng = np.random.default_rng(seed)
    dims = np.array([119, 4, 12, 12, 10, 6, 6, 2, 2])
    nf = (rng.integers(0, 1 << 30, (n, 9)) % dims).astype(np.int32)
    ring = np.stack([np.arange(n), (np.arange(n) + 1) % n], 1)
    el = np.concatenate([ring, rng.integers(0, n, (m - n, 2))]).astype(np.int32)
    ea = np.stack([rng.integers(0, 5, m), rng.integers(0, 6, m), rng.integers(0, 2, m)], 1).astype(np.int32)
    return gp.GraphBatch(np.array([n], np.int32), np.array([m], np.int32), nf, el, ea)


def _shape_batch(shape):
    from flowgnn_amd import graphpack as gp
    if shape == "hep10k":      # kNN graphs: every node sums 16 messages (the case that wraps on the FPGA)
        return gp.synth_hep10k_batch(96, seed=71, with_eigen=False)
    if shape == "caps":        # the reference's MAX_NODE / MAX_EDGE, next to ordinary molecules
        mol = gp.synth_molhiv_batch(40, seed=72)
        return gp.concat_batches([mol.slice(0, 20), _big_graph(500, 5500, 73), mol.slice(20, 40), _big_graph(183, 378, 74)])
    if shape == "molpcba":     # a full-size sample of GCN's own dataset shape
        return gp.synth_molpcba_batch(4096, seed=75)
    raise ValueError(shape)


# measured on MI355X (the flag trips where the oracle's activations say it must; trained GIN on kNN graphs grows ~6x per layer)
EXPECT_RERUN = {("GIN", "hep10k"): True, ("GIN", "caps"): False, ("GIN-VN", "hep10k"): True, ("GIN-VN", "caps"): True,
                ("GCN", "molpcba"): False, ("GCN", "caps"): False}


@pytest.mark.gpu
@pytest.mark.parametrize("per_layer", [False, True], ids=["resident", "per-layer"])
@pytest.mark.parametrize("model,shape", [("GIN", "hep10k"), ("GIN", "caps"), ("GIN-VN", "hep10k"), ("GIN-VN", "caps"), ("GCN", "molpcba"), ("GCN", "caps")])
def test_trained_weights_on_other_graph_shapes(model, shape, per_layer, oracle):
    """Trained GIN / GIN-VN on dense (degree-16) graphs and on graphs at the reference's caps, trained GCN on a full-size molpcba
    sample.  Trained GIN grows ~6x per layer on degree-16 graphs (h_5 up to 2.7e4, GIN-VN 1.7e5; the reference's Q6.10 wraps there,
    SURVEY 0.1), so some of these batches DO leave the split-f16 range: the flag must then trip and the fp32 re-run must match the
    oracle; exact_reruns is asserted either way (EXPECT_RERUN)."""
    from flowgnn_amd import Engine, graphpack as gp
    base = model.replace("-VN", "")
    w = trained(base)
    b = _shape_batch(shape)
    if model == "GIN-VN":
        b = gp.add_virtual_nodes(b)
    fwd = getattr(oracle, base.lower() + "_forward")
    want, hd = fwd(b, [w], dump_h=True, nthreads=8)
    assert np.isfinite(want).all()
    e = Engine(model, device=0, options={SWITCH[base]: 0} if per_layer else {})
    try:
        e.set_weights(w)
        got = e.forward(b)
        reruns = e.exact_reruns()
    finally:
        e.close()
    hmax = [float(np.abs(np.asarray(h)).max()) for h in hd]
    scale = max(1.0, max(hmax))
    assert_close(got, want, scale, what=(model, shape))
    # The range flag, either way.  h_1..h_4 are operands of the next layer's products (a = h + sum of ReLU'd terms >= h): beyond 6e4
    # the split-f16 kernels MUST have raised the flag and the engine repeated the pass on the fp32 pipe; where every activation
    # (aggregates and hidden units included: bounded here by 64 x the largest row entry) stays below it, they must NOT have.
    if max(hmax[1:5]) > 6.0e4:
        assert reruns >= 1, (model, shape, hmax)
    elif 64.0 * max(hmax) < 6.0e4:
        assert reruns == 0, (model, shape, hmax)
    assert EXPECT_RERUN[(model, shape)] == (reruns > 0), (model, shape, reruns, hmax)


def test_q_oracle_is_in_the_range_of_the_three_logits_the_reference_itself_produced(oracle):
    """The only outputs of the reference's own code on record: SURVEY.md section 8(c) -- its eight GIN sources compiled against a throw-away
    ap_fixed stand-in (not shipped, not allowed here) with the real GIN/*.bin weights gave -3.95898438 (19 nodes / 40 edges), -3.96875
    (6 / 12) and -3.90429688 (25 / 52) on synthetic chain + ring molecules, all multiples of 2^-10.  The molecules' features were not
    recorded, so this pins no bit: it checks that the fixed-point oracle, on chain + ring molecules of those sizes with the shipped
    weights, produces multiples of 2^-10 in the same narrow band (a wrong weight offset, a transposed matrix or a missing ReLU moves
    trained GIN logits by whole units)."""
    from flowgnn_amd import graphpack as gp
    w = trained("GIN")
    recorded = {(19, 40): -3.95898438, (6, 12): -3.96875, (25, 52): -3.90429688}
    got = []
    for feat in ([0] * 9, [1] * 9, [5, 0, 4, 5, 3, 0, 2, 0, 0]):
        for attr in ([0, 0, 0], [0, 0, 1]):
            for (n, e), _ in recorded.items():
                und = [(i, i + 1) for i in range(n - 1)] + [(0, n - 1)] + ([(0, n // 2)] if e // 2 - (n - 1) == 2 else [])
                el = np.array([p for a, b in und for p in ((a, b), (b, a))], np.int32)
                assert el.shape[0] == e
                b = gp.GraphBatch(np.array([n], np.int32), np.array([e], np.int32), np.tile(np.array(feat, np.int32), (n, 1)), el,
                                  np.tile(np.array(attr, np.int32), (e, 1)))
                got.append(float(np.ravel(oracle.gin_forward_q(b, [w]))[0]))
    got = np.array(got)
    assert np.array_equal(got * 1024, np.round(got * 1024))  # Q6.10 patterns
    lo, hi = min(recorded.values()), max(recorded.values())
    assert got.min() > lo - 0.75 and got.max() < hi + 0.75, (got.min(), got.max())
    assert abs(np.median(got) - np.median(list(recorded.values()))) < 0.5, np.median(got)

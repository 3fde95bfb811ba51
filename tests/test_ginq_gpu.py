"""GPU parity for the Q6.10 mode of GIN / GIN-VN (flowgnn_set_numeric_mode): the HIP kernels of
ginq.hip against the C oracle oracle/ginq_oracle.c.  Integer arithmetic on 16-bit patterns: BIT-EXACT, whatever the
batch composition or order."""
import numpy as np
import pytest

from flowgnn_amd import Engine, FlowGNNError, graphpack as gp, weights

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def qeng(gin_weights):
    e = Engine("GIN", device=0)
    e.set_weights(gin_weights)
    e.set_numeric_mode("q6.10")
    yield e
    e.close()


def patterns(out):
    p = np.rint(out.astype(np.float64) * 1024.0)
    assert np.array_equal(p / 1024.0, out.astype(np.float64))  # outputs are exact multiples of 2^-10
    return p.astype(np.int64)


def test_bit_exact_vs_oracle(qeng, oracle, gin_weights):
    for b in (gp.synth_molhiv_batch(300, seed=5), gp.synth_hep10k_batch(12, seed=9),      # kNN: the sums really wrap
              gp.add_virtual_nodes(gp.synth_molhiv_batch(64, seed=6))):                    # GIN-VN hubs
        got = qeng.forward(b)
        want_f, want_q = oracle.gin_forward_q(b, [gin_weights], nthreads=8)
        assert np.array_equal(patterns(got), want_q.astype(np.int64))
        assert np.array_equal(got, want_f)


def test_order_and_split_invariance(qeng):
    b = gp.synth_molhiv_batch(500, seed=77)
    out = qeng.forward(b)
    perm = np.random.default_rng(1).permutation(500)
    shuffled = gp.concat_batches([b.slice(int(g), int(g) + 1) for g in perm])
    assert np.array_equal(qeng.forward(shuffled), out[perm])
    assert np.array_equal(qeng.forward(b.slice(100, 260)), out[100:260])


def test_mode_switch_and_unknown_mode(oracle, gin_weights):
    b = gp.synth_molhiv_batch(40, seed=2)
    e = Engine("GIN", device=0)
    e.set_weights(gin_weights)
    f32 = e.forward(b)
    e.set_numeric_mode("q6.10")
    q = e.forward(b)
    e.set_numeric_mode("f32")
    assert np.array_equal(e.forward(b), f32) and not np.array_equal(q, f32)
    assert np.allclose(f32, oracle.gin_forward(b, [gin_weights]), rtol=1e-4, atol=1e-4)
    e.close()
    g = Engine("GCN", device=0)  # every model has its fixed-point mode (tests/test_modelq_gpu.py); an unknown mode id is refused
    g.set_numeric_mode("q6.10")
    assert g.lib.flowgnn_set_numeric_mode(g._h, 7) == 8
    g.close()

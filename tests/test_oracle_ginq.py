"""CPU tests of the Q6.10 GIN oracle (oracle/ginq_oracle.c): the C restatement against an independent
vectorised NumPy restatement (bit-exact: integer arithmetic mod 2^16), its quantisation primitives, and how far it is
from the float oracle (truncation noise; the reference measured mean +0.03, sigma 0.14 on its own graphs)."""
import numpy as np

from flowgnn_amd import graphpack as gp, weights
from tests import numpy_ref


def test_quantisation_primitives(oracle):
    import ctypes as C
    lib = oracle.load()
    lib.orc_q16_from_float.restype = C.c_int16
    lib.orc_q16_from_float.argtypes = [C.c_float]
    q = lambda x: lib.orc_q16_from_float(x)
    assert q(0.0) == 0 and q(1.0) == 1024 and q(-1.0) == -1024
    assert q(0.00097) == 0 and q(-0.00001) == -1          # truncation toward -inf, not toward zero
    assert q(31.999) == 32766 and q(32.0) == -32768       # wrap, not saturation
    assert q(-32.0) == -32768 and q(-32.001) == 32766


def test_c_oracle_matches_numpy_bit_exact(oracle, gin_weights):
    for b in (gp.synth_molhiv_batch(40, seed=11), gp.synth_hep10k_batch(6, seed=2)):  # kNN graphs: sums wrap for real
        out, out_q, hd = oracle.gin_forward_q(b, [gin_weights], dump_h=True)
        ref_q, hs = numpy_ref.gin_forward_q(b, gin_weights, return_h=True)
        assert np.array_equal(hd.astype(np.int64), hs)
        assert np.array_equal(out_q.astype(np.int64), ref_q)
        assert np.array_equal(out, out_q.astype(np.float32) / np.float32(1024))


def test_threads_and_weight_sets(oracle, gin_weights):
    b = gp.synth_molhiv_batch(30, seed=3)
    w2 = weights.synth_gin_weights(seed=8)
    rw = np.zeros(30, np.int32); rw[0] = 1; rw[17] = 1
    a = oracle.gin_forward_q(b, [gin_weights, w2], reload_weights=rw, nthreads=1)[1]
    c = oracle.gin_forward_q(b, [gin_weights, w2], reload_weights=rw, nthreads=4)[1]
    assert np.array_equal(a, c)
    assert np.array_equal(a[:17], oracle.gin_forward_q(b.slice(0, 17), [gin_weights])[1])
    assert np.array_equal(a[17:], oracle.gin_forward_q(b.slice(17, 30), [w2])[1])


def test_distance_from_float_oracle(oracle, gin_weights):
    """Where the two number systems can be compared exactly: the atom encoder adds nine quantised table rows, so
    0 <= float - Q < 9 * 2^-10 element-wise.  Deeper stages accumulate one truncation per stored product (up to 2^-10
    each, 100-200 per output) and are amplified by the layers; with the shipped weights (when the reference checkout is
    present) the logits of graphs that never wrap stay within ~1.5 of the float oracle (mean +0.03, sigma 0.14,
    max 0.75 on the reference's own test graphs)."""
    import os
    b = gp.synth_molhiv_batch(120, seed=21)
    f, hd = oracle.gin_forward(b, [gin_weights], dump_h=True, nthreads=4)
    q, _, hq = oracle.gin_forward_q(b, [gin_weights], dump_h=True, nthreads=4)
    d0 = hd[0].astype(np.float64) - hq[0].astype(np.float64) / 1024.0
    assert d0.min() >= -1e-6 and d0.max() < 9 / 1024 + 1e-6, (d0.min(), d0.max())
    assert np.isfinite(q).all() and np.abs(q).max() < 32.0
    if os.path.exists("/root/reference/GIN/gin_ep1_mlp_1_weights_dim100.bin"):
        wr = weights.LOADERS["GIN"]("/root/reference/GIN")
        f, hd = oracle.gin_forward(b, [wr], dump_h=True, nthreads=4)
        q = oracle.gin_forward_q(b, [wr], nthreads=4)[0]
        off = b.node_offsets()
        calm = np.array([np.abs(hd[:, off[g]:off[g + 1]]).max() < 10.0 for g in range(b.num_graphs)])
        d = (q - f)[calm]
        assert calm.sum() > 30 and np.abs(d).mean() < 0.8 and np.abs(d).max() < 4.0, (calm.sum(), np.abs(d).mean(), np.abs(d).max())

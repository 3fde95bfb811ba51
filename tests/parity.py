"""The float tolerance of the parity tests, in one place: |gpu - oracle| <= REL * (scale + |oracle|),
scale = max(1, the oracle's own largest magnitude) -- the activations' when the oracle dumped them (intermediate sums are that large,
so that is the size of one fp32 rounding), otherwise the expected values'.  No literal scales: what a comparison is relative to is
measured on the oracle's output each time.  REL = 1e-4 for every model; a test that needs more says why beside its call."""
import numpy as np

REL = 1e-4


def oracle_scale(*arrays):
    return max([1.0] + [float(np.abs(np.asarray(a)).max()) for a in arrays if np.asarray(a).size])


def err_ratio(got, want, scale=None, rel=REL):
    """max over elements of |got - want| / (rel * (scale + |want|)): <= 1 passes."""
    got, want = np.asarray(got, np.float64), np.asarray(want, np.float64)
    s = oracle_scale(want) if scale is None else max(1.0, float(scale))
    if got.size == 0:
        return 0.0
    return float((np.abs(got - want) / (rel * (s + np.abs(want)))).max())


def assert_close(got, want, scale=None, rel=REL, what=""):
    got, want = np.asarray(got), np.asarray(want)
    assert got.shape == want.shape, (what, got.shape, want.shape)
    assert np.isfinite(got).all(), (what, "non-finite values on the GPU side")
    r = err_ratio(got, want, scale, rel)
    assert r <= 1.0, (what, f"max |gpu - oracle| = {float(np.abs(got.astype(np.float64) - want).max()):.3e}",
                      f"{r:.2f} x the bound (rel {rel:g}, scale {oracle_scale(want) if scale is None else scale:.3g})")

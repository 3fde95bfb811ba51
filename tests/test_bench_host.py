"""Host-side pieces of bench.py that need no GPU: the CPU count the cpu_baseline leg is sized from, and the model table."""
import builtins
import io

import bench


def _fake_open(files):
    real = builtins.open

    def opener(path, *a, **k):
        if path in files:
            if files[path] is None:
                raise OSError(path)
            return io.StringIO(files[path])
        return real(path, *a, **k)
    return opener


def test_effective_cpus_follows_the_cgroup_quota(monkeypatch):
    monkeypatch.setattr(bench.os, "sched_getaffinity", lambda pid: set(range(256)), raising=False)
    monkeypatch.setattr(builtins, "open", _fake_open({"/sys/fs/cgroup/cpu.max": "1600000 100000\n"}))
    assert bench.effective_cpus() == 16  # the GPU pool's boxes: 256 hardware threads under a 16-CPU quota
    monkeypatch.setattr(builtins, "open", _fake_open({"/sys/fs/cgroup/cpu.max": "max 100000\n"}))
    assert bench.effective_cpus() == 256
    monkeypatch.setattr(builtins, "open", _fake_open({"/sys/fs/cgroup/cpu.max": "150000 100000\n"}))
    assert bench.effective_cpus() == 2  # a fractional quota rounds up
    monkeypatch.setattr(builtins, "open", _fake_open({"/sys/fs/cgroup/cpu.max": None,
                                                      "/sys/fs/cgroup/cpu/cpu.cfs_quota_us": "800000\n",
                                                      "/sys/fs/cgroup/cpu/cpu.cfs_period_us": "100000\n"}))
    assert bench.effective_cpus() == 8  # cgroup v1
    monkeypatch.setattr(builtins, "open", _fake_open({"/sys/fs/cgroup/cpu.max": None, "/sys/fs/cgroup/cpu/cpu.cfs_quota_us": None}))
    assert bench.effective_cpus() == 256


def test_model_table_is_complete():
    for name, m in bench.MODELS.items():
        assert {"metric", "dataset", "graphs", "agg_bytes", "flops", "hbm_kernels", "mfma_kernels", "workload"} <= set(m), name
        n, e = 1000, 2200
        assert m["agg_bytes"](n, e) > 0 and m["flops"](n, e) > 0
        fb = m.get("fused_bytes")
        if isinstance(fb, dict):
            assert set(fb) <= set(m["mfma_kernels"])
            assert all(f(n, e) > 0 for f in fb.values())
        elif fb is not None:
            assert fb(n, e) > 0

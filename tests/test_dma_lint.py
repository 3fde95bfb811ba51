"""The LDS-DMA ordering rule on the assembly hipcc produces (scripts/dev/dma_lint.py; round-4 advisor + verdict item 2).

hipcc neither knows that lds_dma16's inline asm writes LDS nor that it counts in vmcnt; the hand-placed `s_waitcnt vmcnt(M)` +
barrier is the only thing that orders a buffer's readers behind its transfer, and where M > 0 (gin_resident_kernel's encoder front
end: `vmcnt(8)` / `vmcnt(6)`, gin_split.hip) the wait covers the transfer only while hipcc keeps at least M younger loads behind it.
This test compiles the kernels' files to gfx950 assembly (no GPU needed), follows every request along every control-flow path to
the wait that covers it, and compares with the committed snapshot: a compiler upgrade or an edit that moves a load across a request
changes the (M, K) list or turns a cover into a carry, and the snapshot must then be regenerated AND re-read:
    bash scripts/dev/regen_dma_lint.sh      (prints what moved; records the hipcc version the snapshot belongs to)"""
import json
import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "scripts", "dev"))
HIPCC = "/opt/rocm/bin/hipcc"
FILES = ["gin_split", "gcn", "gat", "pna", "dgn"]


def hipcc_version():
    out = subprocess.run([HIPCC, "--version"], capture_output=True, text=True).stdout
    return " | ".join(l.strip() for l in out.splitlines()[:2])


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not installed")
def test_every_lds_dma_request_is_covered_as_in_the_snapshot(tmp_path):
    import dma_lint
    src = os.path.join(ROOT, "flowgnn_amd", "csrc")

    def asm(f):
        out = str(tmp_path / (f + ".s"))
        subprocess.run([HIPCC, "-O3", "-std=c++17", "--offload-arch=gfx950", "--cuda-device-only", "-S", os.path.join(src, f + ".hip"), "-o", out],
                       check=True, capture_output=True)
        return out

    with ThreadPoolExecutor(len(FILES)) as ex:
        paths = list(ex.map(asm, FILES))
    if os.environ.get("KEEP_DMA_LINT_ASM"):  # for regenerating the snapshot (docstring)
        shutil.copytree(tmp_path, "/tmp/dma_lint", dirs_exist_ok=True)
    got = {os.path.basename(p): json.loads(json.dumps(dma_lint.snapshot(p))) for p in paths}
    want = json.load(open(os.path.join(ROOT, "tests", "golden", "dma_lint.json")))
    # the rule itself, whatever the snapshot says (the hard failure): every request meets a covering wait on every path that does not
    # end the wave
    for f, ks in got.items():
        for k, r in ks.items():
            assert r["covers_vmcnt0"] + len(r["counted"]) > 0, (f, k)
            assert all(K >= M for M, K in r["counted"]), (f, k, r["counted"])
    # the (M, K) snapshot is a property of ONE compiler's schedule: compared only under the hipcc that made it (another ROCm moves loads
    # around harmlessly; the rule above still holds it to account)
    made_by = want.pop("_hipcc", None)
    here = hipcc_version()
    if made_by is not None and made_by != here:
        import warnings
        warnings.warn(f"tests/golden/dma_lint.json was made by hipcc {made_by!r}, this is {here!r}: snapshot not compared (rule assertions passed)")
        return
    assert got == want, "LDS-DMA coverage moved: re-read the kernels' waits, then regenerate tests/golden/dma_lint.json (scripts/dev/regen_dma_lint.sh)"

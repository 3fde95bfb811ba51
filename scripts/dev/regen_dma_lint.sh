#!/bin/bash
# Regenerate tests/golden/dma_lint.json after a kernel edit (from the repo root), printing what moved: re-read the waits of every kernel
# it names before committing the new snapshot (tests/test_dma_lint.py, scripts/dev/dma_lint.py).
set -e
KEEP_DMA_LINT_ASM=1 python -m pytest tests/test_dma_lint.py -x -q >/dev/null 2>&1 || true
python scripts/dev/dma_lint.py --json /tmp/dma_lint/*.s > /tmp/dma_lint_new.json
python - <<'PY'
import json
a = json.load(open("tests/golden/dma_lint.json")); b = json.load(open("/tmp/dma_lint_new.json"))
a.pop("_hipcc", None)
for f in sorted(set(a) | set(b)):
    for k in sorted(set(a.get(f, {})) | set(b.get(f, {}))):
        if a.get(f, {}).get(k) != b.get(f, {}).get(k):
            print(f, k[:90]); print("   old", a.get(f, {}).get(k)); print("   new", b.get(f, {}).get(k))
PY
python - <<'PY'
import json, subprocess
d = json.load(open("/tmp/dma_lint_new.json"))
out = subprocess.run(["/opt/rocm/bin/hipcc", "--version"], capture_output=True, text=True).stdout
d["_hipcc"] = " | ".join(l.strip() for l in out.splitlines()[:2])  # the snapshot is compared only under this compiler (tests/test_dma_lint.py)
json.dump(d, open("tests/golden/dma_lint.json", "w"), indent=1, sort_keys=True)
PY
python -m pytest tests/test_dma_lint.py -q | tail -1

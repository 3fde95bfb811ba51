#!/bin/bash
# Regenerate tests/golden/dma_lint.json after a kernel edit (from the repo root), printing what moved: re-read the waits of every kernel
# it names before committing the new snapshot (tests/test_dma_lint.py, scripts/dev/dma_lint.py).
set -e
KEEP_DMA_LINT_ASM=1 python -m pytest tests/test_dma_lint.py -x -q >/dev/null 2>&1 || true
python scripts/dev/dma_lint.py --json /tmp/dma_lint/*.s > /tmp/dma_lint_new.json
python - <<'PY'
import json
a = json.load(open("tests/golden/dma_lint.json")); b = json.load(open("/tmp/dma_lint_new.json"))
for f in sorted(set(a) | set(b)):
    for k in sorted(set(a.get(f, {})) | set(b.get(f, {}))):
        if a.get(f, {}).get(k) != b.get(f, {}).get(k):
            print(f, k[:90]); print("   old", a.get(f, {}).get(k)); print("   new", b.get(f, {}).get(k))
PY
cp /tmp/dma_lint_new.json tests/golden/dma_lint.json
python -m pytest tests/test_dma_lint.py -q | tail -1

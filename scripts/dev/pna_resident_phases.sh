#!/bin/bash
# Phase prices of pna_resident_kernel: the shipped library against its TIMING variants (-DFLOWGNN_DEV -DPNAR_TIMING=<bits>; wrong results
# on purpose) on one box.  Build the variants first, from the repo root:
#   for b in 1 2 4 8 12; do bash scripts/dev/variant.sh pnat$b pna.hip "-DFLOWGNN_DEV -DPNAR_TIMING=$b"; done
G=${1:-65536}
for v in base 1 2 4 8 12; do
  lib=scripts/dev/_pnat$v.so; [ $v = base ] && lib=flowgnn_amd/libflowgnn_hip.so
  [ -f $lib ] || continue
  echo "variant $v: $(python scripts/dev/ab.py PNA $G $lib $lib 1 2>/dev/null | head -1)"
done

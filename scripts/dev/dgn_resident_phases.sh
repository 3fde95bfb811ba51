#!/bin/bash
# Phase prices of dgn_resident_kernel: the shipped library against its TIMING variants (csrc/dev/dgn_timing_variants.h; wrong results on
# purpose) on one box.  Build the variants first, from the repo root:
#   for b in 1 2 3 4 8 12 16 32 47; do bash scripts/dev/variant.sh dgnt$b dgn.hip "-DFLOWGNN_DEV -DDGNR_TIMING=$b"; done
G=${1:-65536}
for v in base 1 2 3 4 8 12 16 32 47; do
  lib=scripts/dev/_dgnt$v.so; [ $v = base ] && lib=flowgnn_amd/libflowgnn_hip.so
  [ -f $lib ] || continue
  echo "variant $v: $(python scripts/dev/ab.py DGN $G $lib $lib 1 | head -1)"
done

#!/bin/bash
# variant.sh NAME FILE.hip "-DFLAG=1 ..."  -> scripts/dev/_NAME.so: the shipped objects with FILE.hip recompiled under the extra flags
# (for scripts/dev/ab.py; *.so is git-ignored).  Run from the repo root after `make -C flowgnn_amd/csrc`.
# The walks' timing variants (csrc/dev/walk_timing_variants.h: GR_CONFLICT_FREE_WALK, GR_ONE_CODE_WALK, GCN_CF_ROWS, GCN_CF_CODES) and the
# ping-pong kernel live behind -DFLOWGNN_DEV: pass it with them, e.g. variant.sh cf gcn.hip "-DFLOWGNN_DEV -DGCN_CF_ROWS".
set -e
name=$1; file=$2; flags=$3
src=flowgnn_amd/csrc
tmp=$(mktemp -d)
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall $flags -c $src/$file -o $tmp/v.o 2>$tmp/log || { grep -A5 "error" $tmp/log | head -60; exit 1; }
objs=""
for o in $src/*.o; do
  if [ "$(basename $o .o)" = "$(basename $file .hip)" ]; then objs="$objs $tmp/v.o"; else objs="$objs $o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o scripts/dev/_$name.so $objs -lpthread
rm -rf $tmp
echo scripts/dev/_$name.so

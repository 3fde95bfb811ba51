#!/bin/bash
# variant.sh NAME FILE.hip "-DFLAG=1 ..."  -> scripts/dev/_NAME.so: the shipped objects with FILE.hip recompiled under the extra flags
# (for scripts/dev/ab.py; *.so is git-ignored).  Run from the repo root after `make -C flowgnn_amd/csrc`.
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

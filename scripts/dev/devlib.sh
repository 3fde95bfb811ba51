#!/bin/bash
# devlib.sh [NAME] ["extra flags"] -> scripts/dev/_NAME.so: the whole library built with -DFLOWGNN_DEV (ablation bits, *_ablate options) in a
# scratch object directory, so the shipped objects under flowgnn_amd/csrc stay as they are.  Run from the repo root.
set -e
name=${1:-dev}; flags=$2
src=flowgnn_amd/csrc
obj=${TMPDIR:-/tmp}/flowgnn_devobj_$name
mkdir -p $obj
pids=""
for f in engine graph_build gin gin_split ginq modelq gcn pna dgn gat; do
  if [ ! -f $obj/$f.o ] || [ $src/$f.hip -nt $obj/$f.o ] || [ -n "$FORCE" ]; then
    ( /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -DFLOWGNN_DEV $flags -c $src/$f.hip -o $obj/$f.o 2>$obj/$f.log || { grep -A5 "error" $obj/$f.log | head -40; exit 1; } ) &
    pids="$pids $!"
  fi
done
for p in $pids; do wait $p; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o scripts/dev/_$name.so $obj/*.o -lpthread
echo scripts/dev/_$name.so

#!/bin/bash
# Experiment driver of the MI355X engine: the counterpart of the reference's run_experiments.sh (run_experiments.sh:9-49,
# 51-128), for a workspace laid out the way the reference lays out its own:
#
#   <root>/<dataset>.zip  (or <root>/graphs/<dataset>.zip)   dataset archives; they unpack to
#   <root>/graphs/graph_info/g<i>_info.txt, <root>/graphs/graph_bin/g<i>_*.bin      the graph pack (GIN/src/host.cc:14-15)
#   <root>/DGN/eig/g<i>.txt                                                         DGN's eigenvectors (DGN/src/host_load.cc:201)
#   <root>/graphs/dataset.txt                                                       name of the pack that is unpacked now
#   <root>/common/includes/dataset/dataset_size.txt                                 its graph count (dataset.hpp)
#   <root>/<MODEL>/*.bin                                                            the model's weight files (<M>/src/host_load.cc)
#
# <root> = $FLOWGNN_ROOT, default: the current directory.  Also accepted: one directory per dataset,
# <root>/graphs/<dataset>/{graph_info,graph_bin,eig,dataset_size.txt}, and weights under <root>/weights/<MODEL>.
#
#   run_experiments.sh all | <dataset> | <model> | <dataset>:<model> ...
#
# Every case runs the whole dataset as one batched launch sequence, NUM_TRIALS times (flowgnn_amd/host), and reports
# "<MODEL> on <dataset>: <ms per graph> ms" -- the figure the reference derives from the profiler's kernel time
# (run_experiments.sh:44-48).  HLS_output.txt is written next to the model's weights, where the reference's host writes it.
set -e
set -o pipefail

SELF_DIR="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"
HOST_BIN="${FLOWGNN_HOST:-$SELF_DIR/flowgnn_amd/host}"
ROOT="${FLOWGNN_ROOT:-$PWD}"
TRIALS="${FLOWGNN_TRIALS:-25}"
known_datasets=(molhiv molpcba hep10k)
known_models=(GIN GIN-VN GCN GAT PNA DGN)
summary=()

say_banner () {
    if [[ -t 1 ]]; then tput bold 2>/dev/null || true; tput setaf 2 2>/dev/null || true; fi
    printf '******* %s *******\n' "$1"
    if [[ -t 1 ]]; then tput sgr0 2>/dev/null || true; fi
}

unpacked_dataset () {  # name of the pack currently unpacked under $ROOT/graphs, if any
    if [[ -f "$ROOT/graphs/dataset.txt" ]]; then tr -d '[:space:]' < "$ROOT/graphs/dataset.txt"; fi
}

# Make $ROOT/graphs hold dataset $1: nothing to do if it is unpacked already or kept in its own directory, otherwise
# drop the previous pack and unpack <dataset>.zip (the archive carries its own dataset.txt / dataset_size.txt).
prepare_dataset () {
    local ds="$1" zip=""
    if [[ -d "$ROOT/graphs/$ds/graph_info" ]]; then return 0; fi
    if [[ "$(unpacked_dataset)" == "$ds" && -d "$ROOT/graphs/graph_info" ]]; then return 0; fi
    for cand in "$ROOT/$ds.zip" "$ROOT/graphs/$ds.zip"; do
        if [[ -f "$cand" ]]; then zip="$cand"; break; fi
    done
    if [[ -z "$zip" ]]; then
        printf 'dataset %s: neither unpacked under %s/graphs nor found as %s.zip\n' "$ds" "$ROOT" "$ds" >&2
        return 1
    fi
    local files
    files="$(unzip -Z1 "$zip" | grep -c -v '/$' || true)"
    printf 'Unpacking dataset %s (%s files) ...\n' "$ds" "$files"
    rm -rf "$ROOT/graphs/graph_bin" "$ROOT/graphs/graph_info" "$ROOT/graphs/dataset.txt" \
           "$ROOT/common/includes/dataset/dataset_size.txt"
    rm -f "$ROOT"/DGN/eig/g*.txt 2>/dev/null || true
    (cd "$ROOT" && unzip -q -o "$zip")
    if [[ ! -f "$ROOT/graphs/dataset.txt" ]]; then printf '%s\n' "$ds" > "$ROOT/graphs/dataset.txt"; fi
}

one_case () {
    local ds="${1,,}" model="${2^^}"
    model="${model//+/-}"
    if [[ " ${known_models[*]} " != *" $model "* ]]; then printf 'Unknown model: %s\n' "$2" >&2; return 1; fi
    say_banner "Running $model on $ds"
    prepare_dataset "$ds"
    local gdir="$ROOT/graphs" eig="$ROOT/DGN/eig" count_file="$ROOT/common/includes/dataset/dataset_size.txt"
    if [[ -d "$ROOT/graphs/$ds/graph_info" ]]; then
        gdir="$ROOT/graphs/$ds"; eig="$gdir/eig"; count_file="$gdir/dataset_size.txt"
    fi
    local wdir="$ROOT/$model"
    if [[ ! -d "$wdir" && -d "$ROOT/weights/$model" ]]; then wdir="$ROOT/weights/$model"; fi
    local args=("$model" --graphs "$gdir" --weights "$wdir" --eig "$eig" --trials "$TRIALS" --out "$wdir/HLS_output.txt")
    if [[ -f "$count_file" ]]; then args+=(--num-graphs "$(tr -d '[:space:]' < "$count_file")"); fi
    local log
    log="$("$HOST_BIN" "${args[@]}")"
    # "<MODEL>: <ms per launch> ms per launch, <ms per graph> ms per graph, ..." (host_main.cpp)
    local per_graph
    per_graph="$(sed -n 's/.* ms per launch, \([0-9.eE+-]*\) ms per graph.*/\1/p' <<< "$log" | tail -n 1)"
    if [[ -z "$per_graph" ]]; then printf '%s\n' "$log" >&2; printf 'no timing line from %s\n' "$HOST_BIN" >&2; return 1; fi
    local line="$model on $ds: $per_graph ms"
    summary+=("$line")
    printf '%s\n\n' "$line"
}

usage () {
    cat <<EOF
Usage: $0 <experiments...>

Experiments:
  all               every model on every dataset
  <dataset>         every model on one dataset
  <model>           one model on every dataset
  <dataset>:<model> one experiment

Datasets: ${known_datasets[*]}
Models:   ${known_models[*]}
Workspace root: \$FLOWGNN_ROOT (default: current directory); trials per case: \$FLOWGNN_TRIALS (default 25)
EOF
}

if [[ "$#" -eq 0 ]]; then usage; exit 1; fi
for a in "$@"; do
    if [[ "$a" == "-h" || "$a" == "--help" ]]; then usage; exit 0; fi
done

for a in "$@"; do
    if [[ "$a" == "all" ]]; then
        for d in "${known_datasets[@]}"; do for m in "${known_models[@]}"; do one_case "$d" "$m"; done; done
    elif [[ "$a" == *:* ]]; then
        one_case "${a%%:*}" "${a#*:}"
    elif [[ " ${known_datasets[*]} " == *" ${a,,} "* ]]; then
        for m in "${known_models[@]}"; do one_case "$a" "$m"; done
    else
        m="${a^^}"; m="${m//+/-}"
        if [[ " ${known_models[*]} " == *" $m "* ]]; then
            for d in "${known_datasets[@]}"; do one_case "$d" "$m"; done
        else
            printf 'Unknown dataset or model: %s\nRun with --help for more information.\n' "$a" >&2
            exit 1
        fi
    fi
done

say_banner "All results"
printf '%s\n' "${summary[@]}"

#!/bin/bash -e
# Counterpart of the reference's run_experiments.sh (run_experiments.sh:28-49,90-115) for the MI355X engine:
#   scripts/run_experiments.sh all | <dataset> | <model> | <dataset>:<model>
# Expects, as the reference does, graph packs extracted under graphs/<dataset>/{graph_info,graph_bin} (plus
# graphs/<dataset>/eig for DGN and graphs/<dataset>/dataset_size.txt) and the models' .bin weights under
# weights/<MODEL>/.  Prints "<MODEL> on <dataset>: <ms per graph> ms" like the reference.
HERE="$(cd "$(dirname "$0")/.." && pwd)"
HOST="$HERE/flowgnn_amd/host"
GRAPHS="${FLOWGNN_GRAPHS:-$HERE/graphs}"
WEIGHTS="${FLOWGNN_WEIGHTS:-$HERE/weights}"
datasets=(molhiv molpcba hep10k)
models=(GIN GIN-VN GCN GAT PNA DGN)
results=()

run_case () {
    dataset="$(tr '[:upper:]' '[:lower:]' <<< "$1")"
    model="$(tr '[:lower:]+' '[:upper:]-' <<< "$2")"
    printf '******* Running %s on %s *******\n' "$model" "$dataset"
    out="$("$HOST" "$model" --graphs "$GRAPHS/$dataset" --weights "$WEIGHTS/$model" --eig "$GRAPHS/$dataset/eig" \
           --out "$HERE/HLS_output.$model.$dataset.txt")"
    ms="$(grep -o '[0-9.]* ms per graph' <<< "$out" | cut -d' ' -f1)"
    results+=("$model on $dataset: $ms ms")
    printf '%s\n\n' "${results[-1]}"
}

arg="${1:-all}"
if [[ "$arg" == "all" ]]; then
    for d in "${datasets[@]}"; do for m in "${models[@]}"; do run_case "$d" "$m"; done; done
elif [[ "$arg" == *:* ]]; then
    run_case "${arg%%:*}" "${arg##*:}"
elif [[ " ${datasets[*]} " == *" $(tr '[:upper:]' '[:lower:]' <<< "$arg") "* ]]; then
    for m in "${models[@]}"; do run_case "$arg" "$m"; done
else
    for d in "${datasets[@]}"; do run_case "$d" "$arg"; done
fi
printf '******* Summary *******\n'
printf '%s\n' "${results[@]}"

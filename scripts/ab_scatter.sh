# dev: the training aggregation's passes under launch-shape variants (usage on the GPU box: bash scripts/ab_scatter.sh)
cd "${GRAFT_REPO_ROOT:?}"
export MDF_TRAIN_STEPS=5 MDF_TRAIN_VERBOSE=1 MDF_TRAIN_STAGE_STREAMS=0
run() {  # label, tab
  out=$(MDF_WARP_TRAIN_TAB=$2 timeout -k 10 120 python3 scripts/bench_train.py 2>&1 | grep -E "^    .*train pass[012]" | awk '{printf "%s ", $(NF-1)}')
  echo "$1: $out"
}
for rep in 1 2; do
run "tab 512" 512
run "tab 256" 256
run "tab 1024" 1024
run "tab 2048" 2048
done

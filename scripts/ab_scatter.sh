# dev: the training aggregation's passes under launch-shape variants (usage on the GPU box: bash scripts/ab_scatter.sh)
cd "${GRAFT_REPO_ROOT:?}"
export MDF_TRAIN_STEPS=5 MDF_TRAIN_VERBOSE=1 MDF_TRAIN_STAGE_STREAMS=0
run() {  # label, blocks
  out=$(MDF_WARP_BWD_BLOCKS=$2 MDF_WARP_BWD_TAB=$3 timeout -k 10 120 python3 scripts/bench_train.py 2>&1 | grep -E "^    .*train pass[3]" | awk '{printf "%s ", $(NF-1)}')
  echo "$1: $out"
}
for rep in 1 2; do
run "bwd blocks 1024" 1024
run "bwd blocks 768" 768
run "bwd blocks 512" 512
run "bwd blocks 256" 256
run "bwd blocks 1536" 1536
run "bwd blocks 512 t1024" 512 1024
run "bwd blocks 512 t2048" 512 2048
done

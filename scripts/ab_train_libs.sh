# usage: bash scripts/ab_train_libs.sh <variant .so> [reps]  -- the cfg3 training step, default library against a variant build, interleaved on one box
cd "${GRAFT_REPO_ROOT:?}"
export MDF_TRAIN_NOPROFILE=1 MDF_TRAIN_STEPS=30 MDF_TRAIN_GRAPH=1
V=$1; reps=${2:-2}
for i in $(seq $reps); do
  echo "variant: $(MDF_HIP_LIB=$V timeout -k 10 120 python3 scripts/bench_train.py 2>&1 | grep 'replay')"
  echo "default: $(timeout -k 10 120 python3 scripts/bench_train.py 2>&1 | grep 'replay')"
done

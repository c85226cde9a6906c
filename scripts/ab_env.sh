# usage: bash scripts/ab_env.sh VAR v1 v2 [reps]   -- interleaved in-box A/B of the cfg3 training step under an environment switch
cd $GRAFT_REPO_ROOT
export MDF_TRAIN_NOPROFILE=1 MDF_TRAIN_STEPS=30
var=$1; a=$2; b=$3; reps=${4:-3}
for i in $(seq $reps); do
  for v in $a $b; do
    export $var=$v
    echo "$var=$v: $(timeout -k 10 120 python3 scripts/bench_train.py 2>&1 | grep 'train step')"
  done
done

cd $GRAFT_REPO_ROOT
export MDF_TRAIN_NOPROFILE=1 MDF_TRAIN_STEPS=30
for nb in 1024 512 768 1024 512 768; do
  export MDF_WGRAD_BLOCKS=$nb
  echo "blocks $nb: $(timeout -k 10 120 python3 scripts/bench_train.py 2>&1 | grep 'train step')"
done

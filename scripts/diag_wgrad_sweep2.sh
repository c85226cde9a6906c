cd $GRAFT_REPO_ROOT
for shape in "64 64 1 12 18 24" "64 64 1 3 18 24" "32 32 1 6 36 48" "32 32 1 24 36 48" "16 16 1 12 72 96" "16 32 1 24 36 48"; do
  for nb in 96 192 384 768 1024; do
    MDF_WGRAD_BLOCKS=$nb timeout -k 5 60 python3 scripts/diag_wgrad_one.py $shape 2>&1 | grep TFLOP
  done
done

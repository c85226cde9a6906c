cd $GRAFT_REPO_ROOT
for shape in "16 16 1 12 72 96" "16 16 1 48 72 96" "32 32 1 24 36 48" "64 64 1 12 18 24"; do
  for th in 1 2 4; do for nb in 512 1024 2048 4096; do
    MDF_WGRAD_TH=$th MDF_WGRAD_BLOCKS=$nb timeout -k 5 60 python3 scripts/diag_wgrad_one.py $shape 2>&1 | grep TFLOP
  done; done
done

# eval headline against the number of views in flight (dev tool; run through gpurun)
cd $GRAFT_REPO_ROOT
for n in 2 3 4 5 6; do
  python bench.py --in-flight $n --steps 100 --warmup 10 --no-cpu-baseline --no-profile --no-training --no-extra-configs 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print($n, d['value'], d['ms_per_step'], d['blocks_ms_per_step']['all'])" || exit 1
done

# after scripts/r05_profiles.sh has run on the GPU box: condense gpurun_out/r5_* into profiles/r05_* (tracked)
set -e
cd "$(dirname "$0")/.."
python scripts/summarize_rocprof.py gpurun_out/r5_prof_eval profiles/r05_bench_cfg2 23 "rocprofv3 --kernel-trace --stats -- python3 bench.py --in-flight 1 --steps 20 --warmup 3 --blocks 1 --no-cpu-baseline --no-profile --no-training --no-extra-configs" forward > /dev/null
python scripts/summarize_rocprof.py gpurun_out/r5_prof_train profiles/r05_train_cfg3 23 "rocprofv3 --kernel-trace --stats -- python3 scripts/bench_train.py  (MDF_TRAIN_STEPS=20, 3 warm-up steps)" "training step" > /dev/null
cp gpurun_out/r5_bench_cfg2.json profiles/r05_bench_cfg2.json
python scripts/summarize_traffic.py gpurun_out/r5_pmc_e_fetch gpurun_out/r5_pmc_e_write 7 profiles/r05_traffic.json profiles/r05_bench_cfg2.json
python scripts/summarize_traffic.py gpurun_out/r5_pmc_t_fetch gpurun_out/r5_pmc_t_write 6 profiles/r05_train_traffic.json

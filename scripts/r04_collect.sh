# after scripts/r04_profiles.sh has run on the GPU box: condense gpurun_out/r4_* into profiles/r04_* (tracked)
set -e
cd "$(dirname "$0")/.."
python scripts/summarize_rocprof.py gpurun_out/r4_prof_eval profiles/r04_bench_cfg2 23 "rocprofv3 --kernel-trace --stats -- python3 bench.py --in-flight 1 --steps 20 --warmup 3 --blocks 1 --no-cpu-baseline --no-profile --no-training --no-extra-configs" forward > /dev/null
python scripts/summarize_rocprof.py gpurun_out/r4_prof_train profiles/r04_train_cfg3 23 "rocprofv3 --kernel-trace --stats -- python3 scripts/bench_train.py  (MDF_TRAIN_STEPS=20, 3 warm-up steps)" "training step" > /dev/null
python scripts/summarize_traffic.py gpurun_out/r4_pmc_e_fetch gpurun_out/r4_pmc_e_write 7 profiles/r04_traffic.json
python scripts/summarize_traffic.py gpurun_out/r4_pmc_t_fetch gpurun_out/r4_pmc_t_write 6 profiles/r04_train_traffic.json

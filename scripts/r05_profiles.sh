# round-5 measurement artefacts (run on the GPU box through gpurun; outputs under gpurun_out/, summaries copied to profiles/ afterwards
# by scripts/r05_collect.sh here).  Every rocprofv3 command profiles `python3 <script>` directly (no env / shell hop behind `--`).
set -e
cd "${GRAFT_REPO_ROOT:?}"
export TMPDIR=/tmp
EVAL="bench.py --in-flight 1 --steps 20 --warmup 3 --blocks 1 --no-cpu-baseline --no-profile --no-training --no-extra-configs"
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r5_prof_eval -- python3 $EVAL > gpurun_out/r5_prof_eval.log 2>&1
echo eval trace done
# (training traces on ONE stream: with the stages' backward chains on their own streams the kernels of three chains run side by side
#  and every per-kernel duration would include the others' share of the chip)
export MDF_TRAIN_NOPROFILE=1 MDF_TRAIN_STEPS=20 MDF_TRAIN_STAGE_STREAMS=0
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r5_prof_train -- python3 scripts/bench_train.py > gpurun_out/r5_prof_train.log 2>&1
echo train trace done
EVALP="bench.py --in-flight 1 --steps 5 --warmup 2 --blocks 1 --no-cpu-baseline --no-profile --no-training --no-extra-configs"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/r5_pmc_e_fetch -- python3 $EVALP > gpurun_out/r5_pmc_e_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/r5_pmc_e_write -- python3 $EVALP > gpurun_out/r5_pmc_e_write.log 2>&1
echo eval pmc done
export MDF_TRAIN_STEPS=3
rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/r5_pmc_t_fetch -- python3 scripts/bench_train.py > gpurun_out/r5_pmc_t_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/r5_pmc_t_write -- python3 scripts/bench_train.py > gpurun_out/r5_pmc_t_write.log 2>&1
echo train pmc done
unset MDF_TRAIN_STAGE_STREAMS
python3 bench.py --steps 20 --warmup 3 > gpurun_out/r5_bench_cfg2.json 2> gpurun_out/r5_bench_cfg2.err
MDF_TRAIN_GRAPH=1 MDF_TRAIN_PIECES=1 MDF_TRAIN_STEPS=30 python3 scripts/bench_train.py > gpurun_out/r5_train_pieces.log 2>&1
echo profiles done

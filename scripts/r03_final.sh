# final artefacts of the round: profiles (scripts/r03_profiles.sh), the default bench line, a 2000-replay soak of the recorded training step
cd $GRAFT_REPO_ROOT
set -e
bash scripts/r03_profiles.sh
python3 bench.py > gpurun_out/r3_bench_final.json 2> gpurun_out/r3_bench_final.err
MDF_TRAIN_GRAPH=1 MDF_TRAIN_NOPROFILE=1 MDF_TRAIN_STEPS=2000 timeout -k 10 200 python3 scripts/bench_train.py > gpurun_out/r3_graph_soak.log 2>&1
echo final done

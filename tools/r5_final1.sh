#!/bin/bash
# Round 5, final evidence, part 1: the GPU suite, the bench lines, the per-rank shard costs.
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out/r5_final
timeout -k 10 900 python3 -m pytest tests -x -q -m gpu > gpurun_out/r5_final/gpu_suite.txt 2>&1; echo "suite rc=$?"; tail -3 gpurun_out/r5_final/gpu_suite.txt
bash tools/bench_round.sh r5 > gpurun_out/r5_final/bench_round.log 2>&1; tail -12 gpurun_out/r5_final/bench_round.log
python3 tools/sim_shard_cost.py 1,2,4,8 > gpurun_out/r5_final/shard_cost.txt 2>&1; grep "^world .:" gpurun_out/r5_final/shard_cost.txt

#!/bin/bash
# The evidence set of a round, in parts that each fit one gpurun call (<= 20 min).  Outputs under gpurun_out/; copy what is to be
# judged into profiles/.  Usage: tools/round_evidence.sh <round tag, e.g. r5> <part>
#   suite     pytest -m gpu, tools/bench_round.sh (every bench line), tools/sim_shard_cost.py 1,2,4,8
#   profiles  tools/profile_round.sh for m256 c1 c3 c4 (rocprofv3 kernel stats + PMC traffic passes, each with the library's identity) + SQ passes for m256 and c4
#   c5        the same profile set for c5 (minutes of rocprofv3 passes of its own)
#   soak      tools/soak.py on m256 / c2 / c4 and the whole GPU suite twice more
#   campaigns tools/campaigns.sh eager many p2 mid
TAG=${1:-r6}; PART=${2:-suite}
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=gpurun_out/${TAG}_final; mkdir -p $O gpurun_out/${TAG}_profiles
sq() {
  for c in "$@"; do
    t=$TAG; [ $c != m256 ] && t=${TAG}_$c
    n=80; [ $c = c4 ] && n=40
    bash tools/pmc_sq.sh $t python3 $R/tools/run_steps.py $c $n > gpurun_out/sq_$t.log 2>&1
    cp gpurun_out/sq_$t/summary_sq.json gpurun_out/${TAG}_profiles/${t}_sq.json 2>/dev/null
    rm -rf gpurun_out/sq_$t/pass* gpurun_out/sq_$t/trace
  done
}
case $PART in
  suite)
    timeout -k 10 900 python3 -m pytest tests -x -q -m gpu > $O/gpu_suite.txt 2>&1; echo "suite rc=$?"; tail -3 $O/gpu_suite.txt
    bash tools/bench_round.sh $TAG > $O/bench_round.log 2>&1; tail -12 $O/bench_round.log
    python3 tools/sim_shard_cost.py 1,2,4,8 > $O/shard_cost.txt 2>&1; grep "^world .:" $O/shard_cost.txt ;;
  profiles) bash tools/profile_round.sh $TAG m256 c1 c3 c4 2>&1 | tail -50; sq m256 c4; ls -la gpurun_out/${TAG}_profiles ;;
  c5) bash tools/profile_round.sh $TAG c5 2>&1 | tail -12; ls -la gpurun_out/${TAG}_profiles ;;
  soak)
    : > gpurun_out/${TAG}_soak.txt
    python3 tools/soak.py m256 30000 2000 >> gpurun_out/${TAG}_soak.txt 2>&1
    python3 tools/soak.py c2 40000 0 >> gpurun_out/${TAG}_soak.txt 2>&1
    python3 tools/soak.py c4 1500 0 >> gpurun_out/${TAG}_soak.txt 2>&1
    cat gpurun_out/${TAG}_soak.txt
    bash tools/repeat_suite.sh 2 gpurun_out/${TAG}_repeat_suite.txt ;;
  campaigns) bash tools/campaigns.sh gpurun_out/${TAG}_campaigns.txt eager many p2 mid ;;
esac

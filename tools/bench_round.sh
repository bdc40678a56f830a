#!/bin/bash
# The round's bench lines on the GPU box, one JSON file each under gpurun_out/<round>_bench/ (copy them to profiles/).
# Usage: tools/bench_round.sh <round tag>
R=${1:-r4}
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/${R}_bench
mkdir -p $OUT
cd $REPO
run() { local name=$1; shift; echo "== $name: $*"; "$@" > $OUT/$name.json 2> $OUT/$name.err; echo "   rc=$? $(head -c 200 $OUT/$name.json)"; }
run ${R}_bench_m256 python3 bench.py
run ${R}_bench_c1 python3 bench.py --config c1 --no-extra --cpu-budget 6
run ${R}_bench_c3 python3 bench.py --config c3 --no-extra --no-cpu
run ${R}_bench_c4 python3 bench.py --config c4 --no-extra --no-cpu --offered-hz 20 --ticks 200
run ${R}_bench_c5 python3 bench.py --config c5 --no-extra --no-cpu --offered-hz 20 --ticks 200
run ${R}_bench_sharded_w1_m256 env GVOM_BENCH_FORCE_SHARDED=1 python3 bench.py --gpus 1 --no-cpu
run ${R}_bench_sharded_w1_c4 env GVOM_BENCH_FORCE_SHARDED=1 python3 bench.py --gpus 1 --config c4 --no-cpu --offered-hz 20 --ticks 100
run ${R}_rehearsal_m256_w2 python3 bench.py --gpus 2 --share-device --no-cpu
run ${R}_rehearsal_c4_w4 python3 bench.py --gpus 4 --share-device --config c4 --no-cpu --offered-hz 20 --ticks 100
run ${R}_rehearsal_c5_w4 python3 bench.py --gpus 4 --share-device --config c5 --no-cpu --offered-hz 20 --ticks 60
ls -la $OUT

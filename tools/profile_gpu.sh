#!/bin/bash
# Runs on the GPU box (via gpurun): rocprofv3 kernel-trace stats + separate PMC passes
# (FETCH_SIZE and WRITE_SIZE cannot share a pass; never combined with trace domains other than
# --kernel-trace).  Usage: tools/profile_gpu.sh <tag> [bench args...]
set -u
TAG=${1:-r1}; shift || true
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
[ -x $REPO/tools/pmc_calib ] || hipcc -O3 --offload-arch=gfx950 -Wno-unused-value $REPO/tools/pmc_calib.hip -o $REPO/tools/pmc_calib
cd /tmp
BENCH="python3 $REPO/bench.py --no-cpu --no-extra --steps 100 --warmup 20 $*"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $BENCH > $OUT/bench_trace.json 2> $OUT/trace.err
for C in FETCH_SIZE WRITE_SIZE TCC_EA0_ATOMIC_sum TCC_HIT_sum TCC_MISS_sum; do
  rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/pmc_$C -- $BENCH > $OUT/bench_$C.json 2> $OUT/pmc_$C.err
done
for C in FETCH_SIZE WRITE_SIZE TCC_EA0_ATOMIC_sum; do
  rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/calib_$C -- $REPO/tools/pmc_calib > $OUT/calib_$C.log 2> $OUT/calib_$C.err
done
find $OUT -name "*.csv" | head -40

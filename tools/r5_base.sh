#!/bin/bash
# Round 5 baseline on today's box: kernel times (rocprofv3) of m256, stage times of m256 / c3 / c4 / c5, the host widening probe.
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r5_base; mkdir -p $O
gcc -O3 -march=native -o /tmp/widen_probe $R/tools/widen_probe.c && /tmp/widen_probe > $O/widen_probe.txt 2>&1
bash $R/tools/prof_kernels.sh > $O/kernels_m256.txt 2>&1
for c in m256 c3 c4 c5; do
  n=300; [ $c = c5 ] && n=60; [ $c = c4 ] && n=150
  python3 $R/tools/run_steps.py $c $n stage > $O/steps_$c.txt 2>&1
done
cat $O/*.txt

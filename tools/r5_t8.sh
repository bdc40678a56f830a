#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r5_t8; mkdir -p $O
cd $R
timeout -k 10 900 python3 -m pytest tests/test_hip_parity.py -x -q -k "golden or c1 or c2_full or metric_grid or c3_ring or c4_at or knobs or interleave or layout or fuzz or random or zero_and_tiny or far_origin or non_finite or strided" > $O/pytest.txt 2>&1
rc=$?
tail -8 $O/pytest.txt
[ $rc -ne 0 ] && exit $rc
for c in m256 c3 c4 c5; do
  n=300; [ $c = c5 ] && n=40; [ $c = c4 ] && n=100
  python3 tools/run_steps.py $c $n stage > $O/steps_$c.txt 2>&1; cat $O/steps_$c.txt
done
bash tools/prof_kernels.sh > $O/kernels_m256.txt 2>&1; cat $O/kernels_m256.txt

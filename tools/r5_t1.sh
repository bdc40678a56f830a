#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r5_t1; mkdir -p $O
cd $R
timeout -k 10 600 python3 -m pytest tests/test_hip_parity.py -x -q -k "eager or golden or one_slot or c1 or c2_full or metric_grid or rejected or fuzz or random or epoch" > $O/pytest.txt 2>&1
rc=$?
tail -15 $O/pytest.txt
[ $rc -ne 0 ] && exit $rc
python3 tools/ab_knob.py m256 300 8 eager=0 eager=1 > $O/ab_m256.txt 2>&1; cat $O/ab_m256.txt
python3 tools/ab_knob.py c2 300 8 eager=0 eager=1 > $O/ab_c2.txt 2>&1; cat $O/ab_c2.txt
bash tools/prof_kernels.sh > $O/kernels_m256.txt 2>&1; cat $O/kernels_m256.txt

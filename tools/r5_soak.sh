#!/bin/bash
# Round 5: soaks of the eager one-slot path and the whole GPU suite twice more in one lease.
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=gpurun_out/r5_soak.txt; : > $O
python3 tools/soak.py m256 30000 2000 >> $O 2>&1
python3 tools/soak.py c2 40000 0 >> $O 2>&1
python3 tools/soak.py c4 1500 0 >> $O 2>&1
cat $O
bash tools/repeat_suite.sh 2 gpurun_out/r5_repeat_suite.txt; cat gpurun_out/r5_repeat_suite.txt

#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r5_t2; mkdir -p $O
cd $R
for c in m256 c2; do
python3 tools/ab_knob.py $c 300 6 eager=1 eager=1,encfuse=4 eager=1,encfuse=2 eager=1,encfuse=16 eager=1,encfuse=20 > $O/ab_$c.txt 2>&1; cat $O/ab_$c.txt
done

#!/bin/bash
# rocprofv3 kernel stats of the step with per-voxel statistics on (GVOM_VOXEL_STATISTICS=1).  Usage: tools/prof_stats.sh <tag> [config]
TAG=${1:-r3}; CFG=${2:-m256}
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/stats_prof_$TAG
rm -rf $OUT; mkdir -p $OUT
for v in 0 1; do echo "== GVOM_VOXEL_STATISTICS=$v"; GVOM_VOXEL_STATISTICS=$v timeout -k 10 100 python3 $REPO/tools/run_steps.py $CFG 300 stage; done
cd /tmp && export TMPDIR=/tmp
GVOM_VOXEL_STATISTICS=1 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 $REPO/tools/run_steps.py $CFG 100 > /dev/null 2>&1
f=$(find $OUT -name "*kernel_stats.csv" | head -1)
cp $f $REPO/gpurun_out/stats_prof_${TAG}_kernel_stats.csv
python3 - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    print("%-28s calls %5s avg %8.1f us  min %8.1f  max %8.1f" % (r["Name"].split("(")[0][:28], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3))
PY

#!/bin/bash
export TMPDIR=/tmp; cd /tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/pf -- python3 $R/tools/run_steps.py m256 200 "$@" > $R/gpurun_out/pf.log 2>&1
f=$(find $R/gpurun_out/pf -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY'
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    if r["Name"].startswith("void k_") or r["Name"].startswith("k_"):
        print("%-40s calls %6s avg %8.2f us" % (r["Name"][:40], r["Calls"], float(r["AverageNs"])/1e3))
PY
rm -rf $R/gpurun_out/pf

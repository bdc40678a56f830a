#!/bin/bash
# SQ counters of k_trace on one rank of a sharded run (one GPU): tools/pmc_shard.sh <world> <rank> "COUNTERS"
REPO=${GRAFT_REPO_ROOT:-/root/repo}
export TMPDIR=/tmp
cd /tmp
for c in $3; do
  out=$REPO/gpurun_out/pmcs_$1_$2_$c
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $out -- python3 $REPO/tools/sim_shard_cost.py m256 $1 $2 > /dev/null 2> $out.err
  f=$(find $out -name "*counter_collection.csv" | head -1)
  python3 - "$f" $c <<'PY'
import csv, sys, collections
f, c = sys.argv[1:3]
acc = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if "k_trace" in r["Kernel_Name"]:
        acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, vals in acc.items():
    print(k, "per launch", sum(vals[2:]) / max(1, len(vals[2:])), "n", len(vals))
PY
done

#!/bin/bash
# PMC comparison of k_trace variants on the GPU box: tools/pmc_trace.sh "6 7" "SQ_WAVES SQ_INSTS_VALU ..."
REPO=${GRAFT_REPO_ROOT:-/root/repo}
export TMPDIR=/tmp
cd /tmp
for v in $1; do
  export GVOM_TRACE_VARIANT=$v
  for c in $2; do
    out=$REPO/gpurun_out/pmc_v${v}_$c
    rocprofv3 --kernel-trace --pmc $c --output-format csv -d $out -- python3 $REPO/bench.py --no-cpu --steps 20 --warmup 5 > /dev/null 2> $out.err
    f=$(find $out -name "*counter_collection.csv" | head -1)
    python3 - "$f" $v $c <<'PY'
import csv, sys, collections
f, v, c = sys.argv[1:4]
acc = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if r["Kernel_Name"].startswith(("void k_trace", "k_trace")):
        acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, vals in acc.items():
    print("variant", v, k, "per launch", sum(vals) / len(vals), "n", len(vals))
PY
  done
done

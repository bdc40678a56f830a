#!/bin/bash
# One rocprofv3 counter pass + a kernel-trace pass over a command, k_trace lines only.  Usage: tools/pmc_quick.sh <tag> <counters...> -- <program> [args]
TAG=$1; shift
CNT=()
while [ "$1" != "--" ]; do CNT+=("$1"); shift; done
shift
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/pq_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --pmc "${CNT[@]}" --output-format csv -d $OUT/pmc -- "$@" > $OUT/pmc.log 2> $OUT/pmc.err
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/pmc/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0]
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in sorted(acc.items()):
    if "k_trace" in k or "k_encode" in k:
        print(k[:60], {c: round(sum(v) / len(v)) for c, v in sorted(d.items())}, "launches", max(len(v) for v in d.values()))
PY

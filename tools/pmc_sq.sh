#!/bin/bash
# SQ / TCC counter passes over one command on the GPU box (via gpurun), summarised per kernel into
# profiles/<tag>_sq.json.  Counters go in their own rocprofv3 runs with --kernel-trace only (never
# with other trace domains).  Usage: tools/pmc_sq.sh <tag> <program> [args...]   (program = python3 ...)
set -u
TAG=$1; shift
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/sq_$TAG
mkdir -p $OUT $REPO/profiles
export TMPDIR=/tmp
cd /tmp
PASSES=(
 "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD"
 "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM"
 "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS_ATOMIC SQ_LDS_ADDR_CONFLICT SQ_THREAD_CYCLES_VALU SQ_INST_CYCLES_VMEM_WR SQ_INSTS_SMEM SQ_INSTS_BRANCH"
 "TCC_EA0_ATOMIC_sum TCC_EA0_WRREQ_sum"
 "GRBM_GUI_ACTIVE"
)
i=0
for P in "${PASSES[@]}"; do
  rocprofv3 --kernel-trace --pmc $P --output-format csv -d $OUT/pass$i -- "$@" > $OUT/pass$i.log 2> $OUT/pass$i.err || echo "pass $i failed" >&2
  i=$((i+1))
done
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- "$@" > $OUT/trace.log 2> $OUT/trace.err
python3 - $OUT $REPO/profiles/${TAG}_sq.json "$*" <<'PY'
import csv, glob, json, os, sys, collections
out, dst, cmd = sys.argv[1:4]
def short(n):
    n = n.replace("void ", "").split("(")[0]
    return n
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.path.join(out, "pass*", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        acc[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
dur = {}
for f in glob.glob(os.path.join(out, "trace", "**", "*kernel_stats.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        dur[short(r["Name"])] = {"calls": int(r["Calls"]), "avg_us": float(r["AverageNs"]) / 1e3}
res = {"command": cmd, "unit": "per launch (mean over launches); SQ_*_CYCLES / SQ_WAIT_* / SQ_ACTIVE_* in quad-cycles",
       "kernels": {}}
for k, cs in acc.items():
    if not k.startswith("k_"):
        continue
    d = {c: sum(v) / len(v) for c, v in cs.items()}
    d.update(dur.get(k, {}))
    if "SQ_INSTS_VALU" in d and d.get("avg_us"):
        # VALU issue bound: 1024 SIMD-32 x 2.4 GHz / 2 cycles per wave64 op = 1.2288e12 wave-instructions/s
        d["valu_issue_bound_us"] = d["SQ_INSTS_VALU"] / 1.2288e12 * 1e6
        d["valu_issue_frac"] = d["valu_issue_bound_us"] / d["avg_us"]
    res["kernels"][k] = d
json.dump(res, open(dst, "w"), indent=1, sort_keys=True)
for k, d in res["kernels"].items():
    print(k, json.dumps({c: (round(v, 1) if isinstance(v, float) else v) for c, v in sorted(d.items())}))
PY

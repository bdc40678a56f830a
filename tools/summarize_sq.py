#!/usr/bin/env python3
"""Summarises the counter passes of tools/pmc_sq.sh (gpurun_out/sq_<tag>/) per kernel into
profiles/<tag>_sq.json.  Usage: tools/summarize_sq.py <pass directory> <output json> <command string>"""
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
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import lib_identity  # noqa: E402
res = {"command": cmd, "library": lib_identity.identity(), "unit": "per launch (mean over launches); SQ_*_CYCLES / SQ_WAIT_* / SQ_ACTIVE_* in quad-cycles",
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

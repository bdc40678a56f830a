#!/usr/bin/env python3
"""Summarises gpurun_out/prof_<tag>/ (written by tools/profile_gpu.sh) into
profiles/<tag>_traffic.json + a markdown table: per kernel average duration (kernel trace),
raw FETCH_SIZE / WRITE_SIZE / TCC_EA0_ATOMIC_sum per launch, the calibration factors measured
with tools/pmc_calib on the same box, and the corrected HBM bytes per launch."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

tag = sys.argv[1] if len(sys.argv) > 1 else "r1"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "gpurun_out", "prof_" + tag)


def short(name):
    n = name.replace("void ", "").split("(")[0]
    return n.split("<")[0] if n.startswith(("k_", "calib_")) else n


def counter_avgs(dirname, counter):
    files = glob.glob(os.path.join(src, dirname, "**", "*counter_collection.csv"), recursive=True)
    acc = defaultdict(list)
    for f in files:
        for r in csv.DictReader(open(f)):
            if r.get("Counter_Name") == counter:
                acc[short(r["Kernel_Name"])].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in acc.items()}, {k: len(v) for k, v in acc.items()}


def trace_avgs():
    files = glob.glob(os.path.join(src, "trace", "**", "*kernel_stats.csv"), recursive=True)
    out = {}
    for f in files:
        for r in csv.DictReader(open(f)):
            # (template instantiations of one kernel -- k_fuse4<2>, k_fuse4<4> -- share a name here: call-weighted mean)
            k, c, a = short(r["Name"]), int(r["Calls"]), float(r["AverageNs"]) / 1e3
            if k in out:
                tot = out[k]["calls"] + c
                out[k] = {"calls": tot, "avg_us": (out[k]["avg_us"] * out[k]["calls"] + a * c) / tot}
            else:
                out[k] = {"calls": c, "avg_us": a}
    return out


GiB = float(1 << 30)
calib = {}
for c in ("FETCH_SIZE", "WRITE_SIZE", "TCC_EA0_ATOMIC_sum"):
    calib[c], _ = counter_avgs("calib_" + c, c)
# counters are in KiB (FETCH/WRITE) ; factor = true bytes / (counter * 1024)
factors = {}
if "calib_read4" in calib["FETCH_SIZE"]:
    factors["read4"] = GiB / (calib["FETCH_SIZE"]["calib_read4"] * 1024)
    factors["read16"] = GiB / (calib["FETCH_SIZE"]["calib_read16"] * 1024)
if "calib_write4" in calib["WRITE_SIZE"]:
    factors["write4"] = GiB / (calib["WRITE_SIZE"]["calib_write4"] * 1024)
    factors["write16"] = GiB / (calib["WRITE_SIZE"]["calib_write16"] * 1024)
if "calib_atomic_scatter" in calib["TCC_EA0_ATOMIC_sum"]:
    factors["atomic_requests_per_scattered_add"] = calib["TCC_EA0_ATOMIC_sum"]["calib_atomic_scatter"] / float(16 << 20)
    factors["atomic_scatter_WRITE_SIZE_KiB"] = calib["WRITE_SIZE"].get("calib_atomic_scatter")
    factors["atomic_scatter_FETCH_SIZE_KiB"] = calib["FETCH_SIZE"].get("calib_atomic_scatter")

trace = trace_avgs()
fetch, _ = counter_avgs("pmc_FETCH_SIZE", "FETCH_SIZE")
write, _ = counter_avgs("pmc_WRITE_SIZE", "WRITE_SIZE")
atom, _ = counter_avgs("pmc_TCC_EA0_ATOMIC_sum", "TCC_EA0_ATOMIC_sum")
hit, _ = counter_avgs("pmc_TCC_HIT_sum", "TCC_HIT_sum")
miss, _ = counter_avgs("pmc_TCC_MISS_sum", "TCC_MISS_sum")
# which calibration applies to which kernel's dominant access width
width = {"k_trace": ("read4", "write4"), "k_encode": ("read16", "write16"), "k_minh": ("read4", "write4"),
         "k_fuse": ("read4", "write4"), "k_fuse4": ("read16", "write16"), "k_fuse1": ("read16", "write16"), "k_encfuse": ("read16", "write16"), "k_map2d": ("read4", "write4")}
kern = {}
for k in sorted(set(fetch) | set(write)):
    if not k.startswith("k_"):
        continue
    rf, wf = width.get(k, ("read4", "write4"))
    fb = fetch.get(k, 0.0) * 1024 * factors.get(rf, 1.0)
    wb = write.get(k, 0.0) * 1024 * factors.get(wf, 1.0)
    kern[k] = {"avg_us": trace.get(k, {}).get("avg_us"), "calls": trace.get(k, {}).get("calls"),
               "FETCH_SIZE_KiB": fetch.get(k), "WRITE_SIZE_KiB": write.get(k),
               "TCC_EA0_ATOMIC_sum": atom.get(k), "TCC_HIT_sum": hit.get(k), "TCC_MISS_sum": miss.get(k),
               "fetch_bytes_corrected": fb, "write_bytes_corrected": wb, "hbm_bytes_corrected": fb + wb}
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import lib_identity  # noqa: E402
out = {"tag": tag, "library": lib_identity.identity(), "calibration_factors": factors, "calibration_raw": calib, "kernels": kern,
       "note": "FETCH_SIZE/WRITE_SIZE in KiB as reported; corrected = raw*1024*factor, factor measured "
               "with tools/pmc_calib (1 GiB known-byte kernels) in the same session"}
os.makedirs(os.path.join(root, "profiles"), exist_ok=True)
json.dump(out, open(os.path.join(root, "profiles", tag + "_traffic.json"), "w"), indent=1)
print(json.dumps(factors, indent=1))
print("| kernel | avg µs | FETCH KiB | WRITE KiB | atomics (EA) | corrected HBM MB |")
print("|---|---|---|---|---|---|")
for k, v in kern.items():
    print("| %s | %s | %s | %s | %s | %.1f |" % (k, v["avg_us"], v["FETCH_SIZE_KiB"], v["WRITE_SIZE_KiB"],
                                            v["TCC_EA0_ATOMIC_sum"], v["hbm_bytes_corrected"] / 1e6))

#!/usr/bin/env python3
"""Per-step wall-time distribution of the headline workload inside one process (fast vs slow processes:
is the difference a constant offset per step or a tail of slow steps?)."""
import os, sys, time, gc
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "g-vom_amd")); sys.path.insert(0, ROOT)
import numpy as np
import gvom, synth, bench
bench.pin_to_gpu_numa(0)
params, scans = synth.config_inputs("m256", n_scans=1)
hip = bench.Hip(); hip.set_device(0)
pc, ego, tf = scans[0]
d = hip.to_device(pc)
g = gvom.Gvom(*params, device=0)
os.environ["GVOM_HOST_TIMING"] = "1"
def step():
    g.process_pointcloud_device(d.value, pc.shape[0], pc.dtype, ego, tf); g.combine_maps()
for _ in range(100): step()
mode = sys.argv[1] if len(sys.argv) > 1 else ""
if "stats" in mode: g.scan_stats()
gc.disable()
n = 2000
ts = np.empty(n + 1)
ts[0] = time.perf_counter()
for k in range(n):
    prof = "prof" in mode and k % 50 == 0
    if prof: g.set_profiling(True)
    step()
    if prof: g.last_stage_ms(); g.set_profiling(False)
    ts[k + 1] = time.perf_counter()
dt = np.diff(ts) * 1e6
print(mode, "cpu %d  mean %.1f  p1 %.1f p10 %.1f p50 %.1f p90 %.1f p99 %.1f max %.1f" % (
    int(open('/proc/self/stat').read().split()[38]), dt.mean(), *np.percentile(dt, [1, 10, 50, 90, 99]), dt.max()))

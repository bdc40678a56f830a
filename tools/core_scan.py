#!/usr/bin/env python3
"""Step time of the headline workload when the calling thread is pinned to single CPUs of the GPU's
NUMA node (and of the other node): is the process-to-process spread of the host gap a per-core effect?"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "g-vom_amd")); sys.path.insert(0, ROOT)
import numpy as np
import gvom, synth, bench
spec = bench.pin_to_gpu_numa(0)
local = sorted(os.sched_getaffinity(0))
params, scans = synth.config_inputs("m256", n_scans=1)
hip = bench.Hip(); hip.set_device(0)
g = gvom.Gvom(*params, device=0)
pc, ego, tf = scans[0]
d = hip.to_device(pc)
def run(k):
    for _ in range(k):
        g.process_pointcloud_device(d.value, pc.shape[0], pc.dtype, ego, tf); g.combine_maps()
run(100)
print("local cpus", spec)
res = []
for rep in range(2):
    for c in local[::max(1, len(local) // 16)]:
        os.sched_setaffinity(0, {c})
        run(20)
        t = time.perf_counter(); run(200); dt = (time.perf_counter() - t) / 200 * 1e6
        res.append((c, dt))
        print("cpu %3d: %.1f us/step" % (c, dt))
os.sched_setaffinity(0, set(local))
run(20); t = time.perf_counter(); run(500); print("whole node: %.1f us/step" % ((time.perf_counter() - t) / 500 * 1e6))

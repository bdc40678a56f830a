#!/usr/bin/env python3
"""Step time over a long run, per 150 steps (does anything drift?).  Usage: tools/step_drift.py <config> <steps> [default|plain]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "g-vom_amd")]
import numpy as np
import bench, gvom, synth
name = sys.argv[1]; steps = int(sys.argv[2]); mode = sys.argv[3] if len(sys.argv) > 3 else "default"
hip = bench.Hip(); hip.set_device(0)
params, scans = synth.config_inputs(name, n_scans=8)
dev = [(hip.to_device(pc), pc.shape[0], pc.dtype, ego, tf) for (pc, ego, tf) in scans]
g = gvom.Gvom(*params) if mode == "default" else gvom.Gvom(*params, voxel_statistics=False)
out = []
t0 = time.perf_counter()
for k in range(steps):
    d, n, dt, ego, tf = dev[k % len(dev)]; g.process_pointcloud_device(d.value, n, dt, ego, tf); o = g.combine_maps()
    if (k + 1) % 150 == 0:
        t1 = time.perf_counter(); out.append((t1 - t0) / 150 * 1e6); t0 = t1
print(name, mode, "us/step per 150 steps:", " ".join("%.0f" % v for v in out), "| interleave", g.get_tuning("interleave"), "dirsort", g.get_tuning("dirsort"))

#!/usr/bin/env python3
"""Wall time of the FIRST steps of a mapper (first-use allocations; statistics on demand: computed for the first three
unread combines, their buffers released at the fourth).  Usage: tools/startup_steps.py [config] [steps] [ondemand|stats|plain]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "g-vom_amd")]
import bench, gvom, synth
name = sys.argv[1] if len(sys.argv) > 1 else "c4"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 12
mode = sys.argv[3] if len(sys.argv) > 3 else "ondemand"
hip = bench.Hip(); hip.set_device(0)
params, scans = synth.config_inputs(name, n_scans=2)
dev = [(hip.to_device(pc), pc.shape[0], pc.dtype, ego, tf) for (pc, ego, tf) in scans]
t0 = time.perf_counter()
g = gvom.Gvom(*params) if mode == "ondemand" else gvom.Gvom(*params, voxel_statistics=(mode == "stats"))
print("%s %s: constructor %.1f ms" % (name, mode, (time.perf_counter() - t0) * 1e3))
ts = []
for k in range(steps):
    d, n, dt, ego, tf = dev[k % 2]
    t0 = time.perf_counter(); g.process_pointcloud_device(d.value, n, dt, ego, tf); t1 = time.perf_counter(); g.combine_maps(); t2 = time.perf_counter()
    ts.append(((t1 - t0) * 1e3, (t2 - t1) * 1e3))
print("  scan / combine ms per step:", " ".join("%.2f/%.2f" % t for t in ts))
print("  total of the first %d steps: %.1f ms" % (steps, sum(a + b for a, b in ts)))

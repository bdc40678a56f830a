#!/usr/bin/env python3
"""Where the host spends a scan + combine step (device-resident cloud): the C-ABI's own timers
(gvom_host_timing) against the wall clock of the Python calls.  Usage: tools/host_gap.py [config] [steps]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "g-vom_amd")]
import numpy as np
import bench, gvom, synth
name = sys.argv[1] if len(sys.argv) > 1 else "m256"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
hip = bench.Hip(); hip.set_device(0)
params, scans = synth.config_inputs(name, n_scans=4)
dev = [(hip.to_device(pc), pc.shape[0], pc.dtype, ego, tf) for (pc, ego, tf) in scans]
g = gvom.Gvom(*params)
for k in range(200):
    d, n, dt, ego, tf = dev[k % 4]; g.process_pointcloud_device(d.value, n, dt, ego, tf); g.combine_maps()
tp = tc = 0.0; acc = {}
t00 = time.perf_counter()
for k in range(steps):
    d, n, dt, ego, tf = dev[k % 4]
    t0 = time.perf_counter(); g.process_pointcloud_device(d.value, n, dt, ego, tf)
    t1 = time.perf_counter(); g.combine_maps()
    t2 = time.perf_counter(); tp += t1 - t0; tc += t2 - t1
    for k_, v in g.host_timing().items(): acc[k_] = acc.get(k_, 0.0) + v
tot = (time.perf_counter() - t00) / steps * 1e6
print("%s: step %.1f us | python: process %.1f combine %.1f loop+timers %.1f | C: %s" % (
    name, tot, tp / steps * 1e6, tc / steps * 1e6, tot - (tp + tc) / steps * 1e6,
    {k_: round(v / steps, 1) for k_, v in acc.items()}))

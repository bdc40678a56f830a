#!/usr/bin/env python3
"""Result-neutral k_trace knobs (gvom_set_tuning) against each other on one config: HIP-event stage times (median over
sampled steps) and the synchronous step time per setting.
usage: tools/knob_sweep.py <config> <steps> <setting> [<setting> ...]      setting = name=value[,name=value...]   ("-" = defaults)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "g-vom_amd")]
import numpy as np
import bench, gvom, synth

name, steps = sys.argv[1], int(sys.argv[2])
hip = bench.Hip(); hip.set_device(0)
params, scans = synth.config_inputs(name, n_scans=4)
dev = [(hip.to_device(pc), pc.shape[0], pc.dtype, ego, tf) for (pc, ego, tf) in scans]
for setting in sys.argv[3:]:
    g = gvom.Gvom(*params)
    if setting != "-":
        for kv in setting.split(","):
            k, v = kv.split("="); g.set_tuning(k, int(v))
    def run(n):
        for k in range(n):
            d, npts, dt, ego, tf = dev[k % 4]; g.process_pointcloud_device(d.value, npts, dt, ego, tf); g.combine_maps()
    run(30)
    t0 = time.perf_counter(); run(steps); dt_us = (time.perf_counter() - t0) / steps * 1e6
    g.set_profiling(True)
    acc = []
    for k in range(60):
        d, npts, dt, ego, tf = dev[k % 4]; g.process_pointcloud_device(d.value, npts, dt, ego, tf); g.combine_maps()
        acc.append(g.last_stage_ms())
    g.set_profiling(False)
    med = {s: float(np.median([a[s] for a in acc])) * 1e3 for s in acc[0]}
    print("%-28s step %7.1f us | trace %6.1f encode %5.1f fuse %5.1f map2d %5.1f" %
          (setting, dt_us, med["trace"], med["encode"], med["fuse"], med["map2d"]), flush=True)
    del g

#!/usr/bin/env python3
"""Sweeps the result-neutral trace knobs (gvom_set_tuning: chunk, ep_row) on one config and prints the
HIP-event stage times (median over sampled steps) and the step time per setting.
Usage: tools/trace_sweep.py [config] [steps] [segs,...] [periods,...]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "g-vom_amd")]
import numpy as np
import bench, gvom, synth

name = sys.argv[1] if len(sys.argv) > 1 else "m256"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 300
hip = bench.Hip(); hip.set_device(0)
params, scans = synth.config_inputs(name, n_scans=4)
dev = [(hip.to_device(pc), pc.shape[0], pc.dtype, ego, tf) for (pc, ego, tf) in scans]
g = gvom.Gvom(*params)
periods = [int(v) for v in (sys.argv[4].split(",") if len(sys.argv) > 4 else ["16"])]
settings = [("segs", int(v), p_) for v in (sys.argv[3].split(",") if len(sys.argv) > 3 else "2,3,4,5,6".split(",")) for p_ in periods]
for kind, chunk, ep in settings:
    g.set_tuning("segs", chunk); g.set_tuning("period", ep)
    for k in range(30):
        d, n, dt, ego, tf = dev[k % 4]; g.process_pointcloud_device(d.value, n, dt, ego, tf); g.combine_maps()
    t0 = time.perf_counter()
    for k in range(steps):
        d, n, dt, ego, tf = dev[k % 4]; g.process_pointcloud_device(d.value, n, dt, ego, tf); g.combine_maps()
    dt_us = (time.perf_counter() - t0) / steps * 1e6
    g.set_profiling(True)
    acc = []
    for k in range(40):
        d, n, dt, ego, tf = dev[k % 4]; g.process_pointcloud_device(d.value, n, dt, ego, tf); g.combine_maps()
        acc.append(g.last_stage_ms())
    g.set_profiling(False)
    med = {s: float(np.median([a[s] for a in acc])) * 1e3 for s in acc[0]}
    print(kind + " %2d period %d: step %.1f us | trace %.1f encode %.1f fuse %.1f map2d %.1f" %
          (chunk, ep, dt_us, med["trace"], med["encode"], med["fuse"], med["map2d"]), flush=True)

#!/usr/bin/env python3
"""A/B of result-neutral knob settings inside ONE process on ONE box: the settings take turns, block by block (boxes differ
by 2-4 %, a fresh process by 1-2 %).  usage: tools/ab_knob.py <config> <steps per block> <blocks> <setting> <setting> ...
setting = name=value[,name=value...] or "-" (defaults)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "g-vom_amd")]
import numpy as np
import bench, gvom, synth

name, steps, blocks = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
settings = sys.argv[4:]
hip = bench.Hip(); hip.set_device(0)
params, scans = synth.config_inputs(name, n_scans=4)
dev = [(hip.to_device(pc), pc.shape[0], pc.dtype, ego, tf) for (pc, ego, tf) in scans]
gs = []
for setting in settings:
    g = gvom.Gvom(*params, voxel_statistics=False)
    if setting != "-":
        for kv in setting.split(","):
            k, v = kv.split("="); g.set_tuning(k, int(v))
    gs.append(g)
def run(g, n):
    for k in range(n):
        d, npts, dt, ego, tf = dev[k % 4]; g.process_pointcloud_device(d.value, npts, dt, ego, tf); g.combine_maps()
for g in gs: run(g, 40)
acc = [[] for _ in gs]
for b in range(blocks):
    for i, g in enumerate(gs):
        t0 = time.perf_counter(); run(g, steps); acc[i].append((time.perf_counter() - t0) / steps * 1e6)
stage = []
for g in gs:                                             # HIP-event time of the kernels (sampled steps: ~80 us longer each)
    g.set_profiling(True); st = []
    for k in range(60):
        d, npts, dt, ego, tf = dev[k % 4]; g.process_pointcloud_device(d.value, npts, dt, ego, tf); g.combine_maps(); st.append(g.last_stage_ms())
    g.set_profiling(False)
    stage.append({k_: round(float(np.median([a[k_] for a in st])) * 1e3, 2) for k_ in ("trace", "encode", "fuse", "map2d")})
for s, a, st in zip(settings, acc, stage):
    print("%-32s median %7.2f us  min %7.2f  (blocks %d x %d steps)  kernels us: %s" % (s, float(np.median(a)), min(a), blocks, steps, st), flush=True)

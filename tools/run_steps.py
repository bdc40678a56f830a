#!/usr/bin/env python3
"""Runs `steps` scan+combine steps of one config (device-resident cloud), for use under rocprofv3.
Usage: tools/run_steps.py [config] [steps] [key=value tuning ...] [stats] [stage] [sync]
(stats: voxel_statistics=True; default: voxel_statistics=False -- the north-star path, no statistics at any step)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "g-vom_amd")]
import bench, gvom, synth
name = sys.argv[1] if len(sys.argv) > 1 else "m256"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 300
hip = bench.Hip(); hip.set_device(0)
poses = ([int(a[6:]) for a in sys.argv[3:] if a.startswith("poses=")] or [4])[0]
params, scans = synth.config_inputs(name, n_scans=poses)
scans = (scans * poses)[:poses]
dev = [(hip.to_device(pc), pc.shape[0], pc.dtype, ego, tf) for (pc, ego, tf) in scans]
g = gvom.Gvom(*params) if "ondemand" in sys.argv[3:] else gvom.Gvom(*params, voxel_statistics=("stats" in sys.argv[3:]))   # ondemand: the class default
stage = "stage" in sys.argv[3:]
for kv in sys.argv[3:]:
    if "=" in kv and not kv.startswith("poses="):
        k, v = kv.split("="); g.set_tuning(k, int(v))
for k in range(60):                                      # first-use allocations (fused compact rows grow to 16 B x V)
    d, n, dt, ego, tf = dev[k % poses]; g.process_pointcloud_device(d.value, n, dt, ego, tf); g.combine_maps()
sync = "sync" in sys.argv[3:]                            # every call followed by a wait for ALL the handle's streams: kernels run alone (intrinsic durations under rocprofv3)
t0 = time.perf_counter()
for k in range(steps):
    d, n, dt, ego, tf = dev[k % poses]; g.process_pointcloud_device(d.value, n, dt, ego, tf)
    if sync: g._lib.gvom_sync(g._h)
    g.combine_maps()
    if sync: g._lib.gvom_sync(g._h)
print("%.1f us/step" % ((time.perf_counter() - t0) / steps * 1e6), end=" ")
if stage:
    import numpy as np
    g.set_profiling(True); acc = []
    for k in range(40):
        d, n, dt, ego, tf = dev[k % poses]; g.process_pointcloud_device(d.value, n, dt, ego, tf); g.combine_maps(); acc.append(g.last_stage_ms())
    print({s_: round(float(np.median([a[s_] for a in acc])) * 1e3, 1) for s_ in acc[0]}, end="")
print()

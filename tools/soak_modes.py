#!/usr/bin/env python3
"""Long runs of the other calling conventions on c2 (host f32 input, float64 + transform, the occupancy form,
per-voxel statistics, four sharded ranks as threads).  Usage: tools/soak_modes.py <host|f64tf|occ|stats|shard> <steps>"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "g-vom_amd"), os.path.join(ROOT, "tests")]
import numpy as np
import gvom, synth
mode = sys.argv[1]; steps = int(sys.argv[2])
params, scans = synth.config_inputs("c2", n_scans=8)
t0 = time.perf_counter()
if mode == "host":
    g = gvom.Gvom(*params)
    for k in range(steps):
        pc, ego, tf = scans[k % 8]; g.process_pointcloud(pc, ego, tf); g.combine_maps()
elif mode == "f64tf":
    g = gvom.Gvom(*params)
    s64 = [(pc.astype(np.float64), ego, np.eye(4)) for pc, ego, tf in scans]
    for k in range(steps):
        pc, ego, tf = s64[k % 8]; g.process_pointcloud(pc, ego, tf); g.combine_maps()
elif mode == "occ":
    g = gvom.Gvom(*params)
    for k in range(steps):
        pc, ego, tf = scans[k % 8]; g.process_pointcloud(pc, ego, tf); g.combine_maps_occupancy()
elif mode == "stats":
    g = gvom.Gvom(*params, voxel_statistics=True)
    for k in range(steps):
        pc, ego, tf = scans[k % 8]; g.process_pointcloud(pc, ego, tf); g.combine_maps()
        if k % 1000 == 0: g.make_debug_voxel_map()
elif mode == "shard":
    from shard_threads import run_ranks
    W = 4
    def body(rank, sh):
        for k in range(steps):
            pc, ego, tf = scans[k % 8]
            share = np.ascontiguousarray(pc[rank::W])
            sh.process_pointcloud(share, ego); sh.combine_maps()
        return True
    run_ranks(W, params, body)
print(mode, steps, "steps ok, %.1f us/step" % ((time.perf_counter() - t0) / steps * 1e6))

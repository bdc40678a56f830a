#!/usr/bin/env python3
"""What is the best composition of a 64-ray bundle for an organised lidar cloud?  The m256 cloud (64 beams x 2048 azimuths, beam-major)
re-ordered on the host so that a bundle holds A consecutive azimuths of B neighbouring beams (A x B = 64), device-resident,
synchronous steps with HIP-event stage times.  (k_trace's results do not depend on the order.)  Usage: tools/bundle_shape_probe.py [config]"""
import os, sys, time, numpy as np
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path[:0] = [R, R + "/g-vom_amd"]
import bench, gvom, synth
name = sys.argv[1] if len(sys.argv) > 1 else "m256"
hip = bench.Hip(); hip.set_device(0)
params, scans = synth.config_inputs(name, n_scans=1)
pc, ego, tf = scans[0]
beams = synth.CONFIGS[name][1]; az = pc.shape[0] // beams


def run(cloud, label):
    g = gvom.Gvom(*params); g.set_tuning("dirsort", -1); g.set_tuning("interleave", 1)
    d = hip.to_device(cloud)
    for k in range(60): g.process_pointcloud_device(d.value, cloud.shape[0], cloud.dtype, ego, tf); g.combine_maps()
    t0 = time.perf_counter()
    for k in range(300): g.process_pointcloud_device(d.value, cloud.shape[0], cloud.dtype, ego, tf); g.combine_maps()
    us = (time.perf_counter() - t0) / 300 * 1e6
    g.set_profiling(True); acc = []
    for k in range(40): g.process_pointcloud_device(d.value, cloud.shape[0], cloud.dtype, ego, tf); g.combine_maps(); acc.append(g.last_stage_ms()["trace"])
    print("%-44s %.1f us/step, trace %.1f us" % (label, us, float(np.median(acc)) * 1e3), flush=True)


grid = pc.reshape(beams, az, 3)
for B in (1, 2, 4, 8, 16, 64):
    A = 64 // B
    if beams % B or az % A: continue
    # bundles: for each group of B beams, for each run of A azimuths: the B x A block, azimuth fastest
    blk = grid.reshape(beams // B, B, az // A, A, 3).transpose(0, 2, 1, 3, 4).reshape(-1, 3)
    run(np.ascontiguousarray(blk), "%2d azimuths x %2d beams per bundle" % (A, B))
# the azimuth-major cloud with the library's own remedy (the probe finds a bundle to be a vertical fan; k_dirbin_* in front of the trace)
g = gvom.Gvom(*params)
cloud = np.ascontiguousarray(grid.transpose(1, 0, 2).reshape(-1, 3)); d = hip.to_device(cloud)
for k in range(60): g.process_pointcloud_device(d.value, cloud.shape[0], cloud.dtype, ego, tf); g.combine_maps()
t0 = time.perf_counter()
for k in range(300): g.process_pointcloud_device(d.value, cloud.shape[0], cloud.dtype, ego, tf); g.combine_maps()
us = (time.perf_counter() - t0) / 300 * 1e6
g.set_profiling(True); acc = []
for k in range(40): g.process_pointcloud_device(d.value, cloud.shape[0], cloud.dtype, ego, tf); g.combine_maps(); acc.append(g.last_stage_ms()["trace"])
print("%-44s %.1f us/step, sort + trace %.1f us (dirsort mode %d)" % (" 1 azimuth  x 64 beams, automatic re-ordering", us, float(np.median(acc)) * 1e3, g.get_tuning("dirsort")), flush=True)

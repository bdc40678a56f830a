#!/usr/bin/env python3
"""bundle_shape_probe.py for the multi-sensor clouds (c4: 4 sensors x 128 beams x 2048 azimuths, sensor-major): a bundle of 64
holds A azimuths x B beams x S sensors (lane order: sensor fastest, then azimuth, then beam), re-ordered on the host, interleave off.
Usage: tools/bundle_shape_probe_multi.py [c4|c5]"""
import os, sys, time, numpy as np
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path[:0] = [R, R + "/g-vom_amd"]
import bench, gvom, synth
name = sys.argv[1] if len(sys.argv) > 1 else "c4"
hip = bench.Hip(); hip.set_device(0)
params, scans = synth.config_inputs(name, n_scans=1)
pc, ego, tf = scans[0]
beams = synth.CONFIGS[name][1]; nsens = synth.SENSORS[name]; az = pc.shape[0] // (beams * nsens)
grid = pc.reshape(nsens, beams, az, 3)


def run(cloud, label, ilv=1, period=0):
    g = gvom.Gvom(*params); g.set_tuning("dirsort", -1); g.set_tuning("interleave", ilv); g.set_tuning("period", period)
    d = hip.to_device(cloud)
    for k in range(12): g.process_pointcloud_device(d.value, cloud.shape[0], cloud.dtype, ego, tf); g.combine_maps()
    g.set_profiling(True); acc = []
    for k in range(12): g.process_pointcloud_device(d.value, cloud.shape[0], cloud.dtype, ego, tf); g.combine_maps(); acc.append(g.last_stage_ms()["trace"])
    print("%-64s trace %.1f us" % (label, float(np.median(acc)) * 1e3), flush=True)


run(pc, "as given, the library's interleave (automatic)", 0)
run(pc, "as given, the library's interleave forced to 8", 8)
run(pc, "as given, the library's interleave forced to 8, period 24", 8, 24)
for A, B, S in ((16, 1, 4), (8, 2, 4), (32, 1, 2), (16, 2, 2), (64, 1, 1), (32, 2, 1), (8, 1, 8), (4, 1, 16), (4, 2, 8)):
    if S > nsens or nsens % S or beams % B or az % A: continue
    # [sensor group][beam group][azimuth run][beam in group][azimuth in run][sensor in group]
    blk = grid.reshape(nsens // S, S, beams // B, B, az // A, A, 3).transpose(0, 2, 4, 3, 5, 1, 6).reshape(-1, 3)
    run(np.ascontiguousarray(blk), "%2d azimuths x %d beams x %2d sensors per bundle" % (A, B, S))
    if S >= 4: run(np.ascontiguousarray(blk), "   ... flush period 24", 1, 24)

#!/usr/bin/env python3
"""Host-side probe: returns ordered by (elevation bin, azimuth sector) seen from the sensor -- elevation fine, azimuth coarse, random
order inside a cell -- against the order given.  For m256 (beam-major lidar), the same cloud azimuth-major, and c1 (random).
Usage: tools/sphere_sort_probe.py"""
import os, sys, time, numpy as np
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path[:0] = [R, R + "/g-vom_amd"]
import bench, gvom, synth
hip = bench.Hip(); hip.set_device(0)


def run(params, cloud, ego, label):
    g = gvom.Gvom(*params); g.set_tuning("dirsort", -1); g.set_tuning("interleave", 1)
    d = hip.to_device(cloud)
    for k in range(40): g.process_pointcloud_device(d.value, cloud.shape[0], cloud.dtype, ego, None); g.combine_maps()
    g.set_profiling(True); acc = []
    for k in range(40): g.process_pointcloud_device(d.value, cloud.shape[0], cloud.dtype, ego, None); g.combine_maps(); acc.append(g.last_stage_ms()["trace"])
    print("%-60s trace %.1f us" % (label, float(np.median(acc)) * 1e3), flush=True)


def sphere_order(pc, ego, res, ne, na, rng):
    d = pc.astype(np.float64) / res - np.array(ego) / res
    az = np.arctan2(d[:, 1], d[:, 0]); el = np.arctan2(d[:, 2], np.hypot(d[:, 0], d[:, 1]))
    ka = np.clip(((az + np.pi) / (2 * np.pi) * na).astype(int), 0, na - 1)
    ke = np.clip(((el + np.pi / 2) / np.pi * ne).astype(int), 0, ne - 1)
    key = ke * na + ka
    return np.lexsort((rng.random(len(key)), key))


rng = np.random.default_rng(0)
for name in ("m256", "c1"):
    params, scans = synth.config_inputs(name, n_scans=1)
    pc, ego, tf = scans[0]
    res = np.array([params[0], params[0], params[1]])
    run(params, pc, ego, name + " as given")
    if name == "m256":
        azm = np.ascontiguousarray(pc.reshape(64, 2048, 3).transpose(1, 0, 2).reshape(-1, 3))
        run(params, azm, ego, name + " azimuth-major (a firing-order cloud)")
    for ne, na in ((256, 32), (512, 32), (256, 16), (1024, 16), (128, 64), (512, 64)):
        o = sphere_order(pc, ego, res, ne, na, rng)
        run(params, np.ascontiguousarray(pc[o]), ego, "%s by (elevation %d bins over 180 deg, azimuth %d sectors)" % (name, ne, na))

# the azimuth-major cloud through a STABLE sort by (sin-elevation row, azimuth sector): inside a cell the input order survives, and
# for a firing-order cloud that is increasing azimuth -- do the beam-major fans come back?
params, scans = synth.config_inputs("m256", n_scans=1)
pc, ego, tf = scans[0]
res = np.array([params[0], params[0], params[1]])
azm = np.ascontiguousarray(pc.reshape(64, 2048, 3).transpose(1, 0, 2).reshape(-1, 3))
d = azm.astype(np.float64) / res - np.array(ego) / res
r = np.linalg.norm(d, axis=1)
for ne, na in ((64, 64), (128, 64), (128, 32), (256, 32), (128, 16), (256, 64)):
    ke = np.clip(((d[:, 2] / r + 1) * 0.5 * ne).astype(int), 0, ne - 1)
    ka = np.clip(((np.arctan2(d[:, 1], d[:, 0]) + np.pi) / (2 * np.pi) * na).astype(int), 0, na - 1)
    o = np.argsort(ke * na + ka, kind="stable")
    run(params, np.ascontiguousarray(azm[o]), ego, "firing order, stable sort by (sin-elevation %d rows, azimuth %d sectors)" % (ne, na))
    # and with the order inside a cell only APPROXIMATELY kept (chunks of 4 returns per cell and block, blocks in shuffled order)
    blk = (np.arange(len(azm)) // 256)
    jit = rng.permutation(blk.max() + 1)[blk]
    o2 = np.lexsort((jit, ke * na + ka))
    run(params, np.ascontiguousarray(azm[o2]), ego, "   ... blocks of 256 returns arriving in random order")

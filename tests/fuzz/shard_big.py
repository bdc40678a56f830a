#!/usr/bin/env python3
"""Slab-sharded handles against the unsharded handle on the WEAK-SCALING clouds of the bench (world x
131,072 returns into one 256 x 256 x 64 map): above three scans' worth of returns sharded handles trace
in 3 segments instead of 6, a path the small fuzz cases never take.  Every rank's rows of the scan
slot and of the fused map must equal the unsharded handle's.  Usage: tests/fuzz/shard_big.py [worlds]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in ("tests", "tests/golden", "g-vom_amd", ""):
    sys.path.insert(0, os.path.join(ROOT, p))
import gvom, synth

def owned(origin_y, xy, r, W):
    sy = (np.arange(xy) + int(origin_y) % xy) % xy
    rows = xy // W
    return (sy >= r * rows) & (sy < (r + 1) * rows)

worlds = [int(w) for w in sys.argv[1].split(",")] if len(sys.argv) > 1 else [4, 8]
params, beams, _ = synth.CONFIGS["c2"]
xy, zs = params[2], params[3]
scene = synth.make_scene(2)
bad = 0
for W in worlds:
    g0 = gvom.Gvom(*params)
    hs = [gvom.Gvom(*params, _shard=(r, W)) for r in range(W)]
    for k in range(3):
        ego = (0.4 * k, -0.3 * k, 0.02 * k)
        full = np.concatenate([synth.lidar_scan(scene, beams=beams, sensor=ego, yaw=2 * np.pi / 2048 * r / W, noise_seed=10 * k + r)
                               for r in range(W)], 0)
        g0.process_pointcloud(full, ego)
        for h in hs:
            h.process_pointcloud(full, ego)
        g0.combine_maps()
        for h in hs:
            h._lib.gvom_combine_fuse(h._h, None); h._lib.gvom_sync(h._h)
        for which in (g0.last_buffer_index, gvom.GVOM_WHICH_FUSED):
            want = g0.read_dense(which)
            for r, h in enumerate(hs):
                got = h.read_dense(which)
                m = np.broadcast_to(owned(got[4][1], xy, r, W)[None, :, None], (zs, xy, xy)).reshape(-1)
                for j, nm in enumerate(("state", "hit", "total", "minh")):
                    a, b = want[j][m], got[j][m]
                    if nm == "state":
                        a, b = np.where(a >= 0, 0, a), np.where(b >= 0, 0, b)
                    if not np.array_equal(a, b):
                        bad += 1
                        print("MISMATCH world %d scan %d rank %d %s %s: %d voxels" % (W, k, r, "slot" if which != gvom.GVOM_WHICH_FUSED else "fused", nm, int(np.sum(a != b))))
    print("world %d: %d returns per scan, 3 scans checked" % (W, full.shape[0]))
print("shard_big: %d mismatches" % bad)
sys.exit(1 if bad else 0)

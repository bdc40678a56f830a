#!/usr/bin/env python3
"""Per-rank kernel cost of the slab-sharded mapper, measured on ONE GPU: for world = 2/4/8, a
sharded handle of every rank r is fed the whole weak-scaling cloud (world x 131,072 points) and
its stage times are printed.  Predicts the multi-GPU critical path (max over ranks) without the
collectives."""
import ctypes
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "g-vom_amd")); sys.path.insert(0, ROOT)
import numpy as np
import gvom, synth

name = sys.argv[1] if len(sys.argv) > 1 else "m256"
worlds = tuple(int(w) for w in sys.argv[2].split(",")) if len(sys.argv) > 2 else (1, 2, 4, 8)
only = tuple(int(r) for r in sys.argv[3].split(",")) if len(sys.argv) > 3 else None
params, beams, _ = synth.CONFIGS[name]
scene = synth.make_scene(2)
for world in worlds:
    clouds = [synth.lidar_scan(scene, beams=beams, yaw=2 * np.pi / 2048 * r / world, noise_seed=r) for r in range(world)]
    full = np.concatenate(clouds, 0)
    worst = {}
    for r in range(world):
        if only is not None and r not in only:
            continue
        g = gvom.Gvom(*params, device=0, _shard=(r, world))
        L = g._lib
        acc, cnt = {}, 0
        for it in range(26):
            g.set_profiling(it >= 6)
            g.process_pointcloud(full, (0.0, 0.0, 0.0))
            L.gvom_combine_fuse(g._h, None)
            L.gvom_sync(g._h)
            if it >= 6:
                for k, v in g.last_stage_ms().items():
                    acc[k] = acc.get(k, 0.0) + v
                cnt += 1
        ms = {k: v / cnt for k, v in acc.items()}
        print("world %d rank %d: trace %.1f encode %.1f fuse %.1f us" % (world, r, ms["trace"] * 1e3, ms["encode"] * 1e3, ms["fuse"] * 1e3))
        for k in ("trace", "encode", "fuse"):
            worst[k] = max(worst.get(k, 0), ms[k] * 1e3)
        del g
    print("world %d critical path: trace %.1f + encode %.1f + fuse %.1f = %.1f us for %d points" % (
        world, worst["trace"], worst["encode"], worst["fuse"], sum(worst.values()), full.shape[0]))

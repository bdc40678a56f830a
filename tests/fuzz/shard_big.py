#!/usr/bin/env python3
"""Sharded map against the unsharded handle at FULL size on one GPU (ranks as threads,
tests/shard_threads.py): every rank contributes its own sensor's scan (weak scaling: world x returns
into ONE map); the rows each rank owns of the scan slot and of the fused map, and every rank's
returned 2-D maps, must equal the unsharded handle's fed with the concatenated cloud.
Usage: tests/fuzz/shard_big.py <config: c2|c4|c5> <worlds, e.g. 4,8> [scans] [buffer] [threads|loopback]
(loopback: the product's communicator over RCCL instead of the thread double; prints the RCCL calls it issued)"""
import os, sys, io, contextlib
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in ("tests", "tests/golden", "g-vom_amd", ""):
    sys.path.insert(0, os.path.join(ROOT, p))
import gvom, synth
from shard_threads import run_ranks

GRIDS = {"c2": (256, 64, 64), "c4": (512, 128, 64), "c5": (1024, 128, 64)}     # xy, z, beams per sensor


def owned(origin_y, xy, r, W):
    sy = (np.arange(xy) + int(origin_y) % xy) % xy
    rows = xy // W
    return (sy >= r * rows) & (sy < (r + 1) * rows)


def main():
    cfg = sys.argv[1] if len(sys.argv) > 1 else "c2"
    worlds = [int(w) for w in sys.argv[2].split(",")] if len(sys.argv) > 2 else [4, 8]
    n_scans = int(sys.argv[3]) if len(sys.argv) > 3 else 3
    buffer = int(sys.argv[4]) if len(sys.argv) > 4 else 1
    transport = sys.argv[5] if len(sys.argv) > 5 and sys.argv[5] != "threads" else None
    xy, zs, beams = GRIDS[cfg]
    params = (0.2, 0.2, xy, zs, buffer) + synth.REF_TAIL
    scene = synth.make_scene(2)
    bad = 0
    for W in worlds:
        per_rank = {"c2": 1, "c4": 4 // min(W, 4) or 1, "c5": max(1, 16 // W)}[cfg]    # sensors per rank
        scans = []
        for k in range(n_scans):
            ego = (0.4 * k, -0.3 * k, 0.02 * k)
            shares = [np.concatenate([synth.lidar_scan(scene, beams=beams if cfg == "c2" else 128, sensor=ego,
                                                       yaw=2 * np.pi / 2048 * (r * per_rank + j) / (W * per_rank),
                                                       noise_seed=100 * k + r * per_rank + j)
                                      for j in range(per_rank)], 0) for r in range(W)]
            scans.append((shares, ego))
        g0 = gvom.Gvom(*params)
        want = []
        for shares, ego in scans:
            g0.process_pointcloud(np.concatenate(shares, 0), ego)
            out = g0.combine_maps()
            want.append((g0.read_dense(g0.last_buffer_index), g0.read_dense(gvom.GVOM_WHICH_FUSED), out, g0.last_buffer_index))
        del g0

        def body(r, sh):
            nbad = 0
            for (shares, ego), (wslot, wfused, wout, b) in zip(scans, want):
                sh.process_pointcloud(shares[r], ego)
                out = sh.combine_maps()
                for a, c in zip(out, wout):
                    if not np.array_equal(a, c):
                        nbad += 1
                for w, which in ((wslot, b), (wfused, gvom.GVOM_WHICH_FUSED)):
                    got = sh.b.g.read_dense(which)
                    m = np.broadcast_to(owned(got[4][1], xy, r, W)[None, :, None], (zs, xy, xy)).reshape(-1)
                    for j, nm in enumerate(("state", "hit", "total", "minh")):
                        a, c = w[j][m], got[j][m]
                        if nm == "state":
                            a, c = np.where(a >= 0, 0, a), np.where(c >= 0, 0, c)
                        if not np.array_equal(a, c):
                            nbad += 1
                            print("MISMATCH world %d rank %d %s %s: %d voxels" % (W, r, "slot" if which != gvom.GVOM_WHICH_FUSED else "fused", nm, int(np.sum(a != c))))
            if transport:
                wire[r] = sh.comm.wire_stats()
            return nbad

        wire = [None] * W
        with contextlib.redirect_stdout(sys.stderr):
            res = run_ranks(W, params, body, transport=transport)
        bad += sum(res)
        if transport:
            print("%s world %d over %s: RCCL calls per rank %s" % (cfg, W, transport, wire))
            if not all(w["p2p_calls"] > 0 and w["allgathers"] == n_scans for w in wire):
                bad += 1
        print("%s world %d: %d returns per scan, %d scans checked" % (cfg, W, sum(s.shape[0] for s in scans[0][0]), n_scans))
    print("shard_big: %d mismatches" % bad)
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()

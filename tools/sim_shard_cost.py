#!/usr/bin/env python3
"""Per-rank cost of the sharded map on ONE GPU (weak scaling: every rank its own 131,072-point sensor,
one 256^3 map).  One functional pass moves every rank's contributions (hipMemcpy) so that all
receive regions hold real data; then each rank's phases are timed ALONE on the GPU:
    local  = gvom_shard_scan_local   (k_trace over its own rays + k_pack + count publish)
    merge  = gvom_shard_scan_merge   (k_unpack of everything it received + k_encode of its rows)
    fuse   = gvom_combine_fuse       (slab fusion + positive-obstacle densities)
    map2d  = gvom_combine_map2d_into (all rows of the 2-D outputs)
and the bytes each rank sends / receives are printed, so the wire time can be budgeted against the
xGMI links (7 x ~50 GB/s per direction).  Usage: tools/sim_shard_cost.py [worlds, e.g. 1,2,4,8] [config]"""
import ctypes, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "g-vom_amd")]
import numpy as np
import bench, gvom, gvom_sharded, synth

worlds = [int(w) for w in sys.argv[1].split(",")] if len(sys.argv) > 1 else [1, 2, 4, 8]
name = sys.argv[2] if len(sys.argv) > 2 else "m256"
params, beams, _ = synth.CONFIGS[name]
scene = synth.make_scene(2)
hip = bench.Hip(); hip.set_device(0)
rt = hip.rt
REPS = 30


def cp(dst, src, n):
    if n:
        assert rt.hipMemcpy(ctypes.c_void_p(dst), ctypes.c_void_p(src), n, 3) == 0


for W in worlds:
    bs = [gvom_sharded.HipShardBackend(params, r, W, 0) for r in range(W)]
    ego = (0.0, 0.0, 0.0)
    shares = []
    for r in range(W):
        pc = synth.lidar_scan(scene, beams=beams, sensor=ego, yaw=2 * np.pi / 2048 * r / W, noise_seed=r)
        shares.append((hip.to_device(pc).value, pc.shape[0], pc.dtype))

    def functional_pass():
        loc = [b.scan_local(shares[r], ego, None) for r, b in enumerate(bs)]
        for b in bs:
            b.sync()
        recv = []
        for me, b in enumerate(bs):
            rq = [loc[s][0][me] if s != me else 0 for s in range(W)]
            re = [loc[s][1][me] if s != me else 0 for s in range(W)]
            b.recv_reserve(re)
            for s in range(W):
                if s != me:
                    cp(b.buffer(3, s)[0], bs[s].buffer(0, me)[0], rq[s] * 4)
                    cp(b.buffer(4, s)[0], bs[s].buffer(1, me)[0], rq[s] * 1024)
                    cp(b.buffer(5, s)[0], bs[s].buffer(2, me)[0], re[s] * 8)
            recv.append((rq, re))
        assert rt.hipDeviceSynchronize() == 0          # (null-stream copies: the handles' streams are not ordered against them)
        for me, b in enumerate(bs):
            b.scan_merge(recv[me][0], recv[me][1], True)
            b.sync()
        return loc, recv

    for _ in range(3):
        loc, recv = functional_pass()
        for b in bs:
            b.combine_fuse(); b.sync()
    rows = []
    for me, b in enumerate(bs):
        t = {"local": 0.0, "merge": 0.0, "fuse": 0.0, "map2d": 0.0}
        for _ in range(REPS):
            t0 = time.perf_counter(); b.scan_local(shares[me], ego, None); b.sync(); t1 = time.perf_counter()
            b.recv_reserve(recv[me][1])
            t2 = time.perf_counter(); b.scan_merge(recv[me][0], recv[me][1], True); b.sync(); t3 = time.perf_counter()
            b.combine_fuse(); b.sync(); t4 = time.perf_counter()
            b.combine_map2d(); t5 = time.perf_counter()
            t["local"] += t1 - t0; t["merge"] += t3 - t2; t["fuse"] += t4 - t3; t["map2d"] += t5 - t4
        us = {k: v / REPS * 1e6 for k, v in t.items()}
        sent = sum(loc[me][0][d] * 1028 + loc[me][1][d] * 8 for d in range(W) if d != me)
        got = sum(recv[me][0]) * 1028 + sum(recv[me][1]) * 8
        rows.append((us, sent, got))
        print("world %d rank %d: local %.1f merge %.1f fuse %.1f map2d %.1f us | sends %.2f MB, receives %.2f MB" %
              (W, me, us["local"], us["merge"], us["fuse"], us["map2d"], sent / 1e6, got / 1e6), flush=True)
    crit = {k: max(r[0][k] for r in rows) for k in rows[0][0]}
    print("world %d: per-rank kernel critical path (max over ranks, host-timed with a sync per phase): %.1f us "
          "= local %.1f + merge %.1f + fuse %.1f + map2d %.1f; max bytes out %.2f MB, in %.2f MB" %
          (W, sum(crit.values()), crit["local"], crit["merge"], crit["fuse"], crit["map2d"],
           max(r[1] for r in rows) / 1e6, max(r[2] for r in rows) / 1e6), flush=True)
    del bs

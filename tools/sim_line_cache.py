#!/usr/bin/env python3
"""Estimates how many 64-B accumulator-line requests k_trace would issue if every wave merged its
updates over windows of K consecutive DDA steps (wave-private LDS line cache) instead of per
step.  Approximate DDA in f64 (statistics only).  Usage: tools/sim_line_cache.py [config]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "g-vom_amd"))
import synth

name = sys.argv[1] if len(sys.argv) > 1 else "m256"
params, scans = synth.config_inputs(name)
xy_res, z_res, xy, zs = params[0], params[1], params[2], params[3]
pc, ego, _ = scans[0]
pc = pc.astype(np.float64)
n = pc.shape[0]
res = np.array([xy_res, xy_res, z_res])
origin = np.floor(np.array(ego) / res) - np.array([xy, xy, zs]) / 2
e = pc / res
p0 = np.array(ego) / res
s = e - p0
rl = np.sqrt((s * s).sum(1))
ok = (pc * pc).sum(1) >= 1.0
sl = s / rl[:, None]
si = np.argmax(np.abs(sl), axis=1)
adom = np.abs(sl[np.arange(n), si])
inc = sl / adom[:, None]
step_len = 1.0 / adom
nsteps = np.where(ok, np.ceil((rl - 1.0) / step_len), 0).astype(int)
nsteps = np.maximum(nsteps, 0)
S = nsteps.max()
print("rays", n, "max steps", S, "total steps", nsteps.sum())
NSEG = 4
seg_len = (S + NSEG - 1) // NSEG

def lines_for(K, shape):
    """distinct (wave, segment, window, line) count; shape = (px, py, pz) patch dims."""
    total_req = 0
    total_heads = 0
    wave = np.arange(n) // 64
    for w0 in range(0, S, 1):
        pass
    keys_all = []
    for k in range(1, S + 1):
        act = nsteps >= k
        if not act.any():
            break
        idx = np.nonzero(act)[0]
        p = p0 + inc[idx] * k
        v = np.floor(p - origin).astype(np.int64)
        ing = (v[:, 0] >= 0) & (v[:, 0] < xy) & (v[:, 1] >= 0) & (v[:, 1] < xy) & (v[:, 2] >= 0) & (v[:, 2] < zs)
        idx = idx[ing]; v = v[ing]
        line = ((v[:, 1] // shape[1]) * 4096 + (v[:, 2] // shape[2])) * 4096 + v[:, 0] // shape[0]
        seg = (k - 1) // seg_len
        win = ((k - 1) % seg_len) // K
        key = ((wave[idx] * NSEG + seg) * 1024 + win) * (1 << 36) + line
        keys_all.append(key)
    keys = np.concatenate(keys_all)
    return keys.size, np.unique(keys).size

for shape in [(4, 4, 1)]:
    for K in [1, 2, 4, 8, 16, 32]:
        upd, req = lines_for(K, shape)
        print("patch", shape, "K", K, "updates", upd, "requests", req)

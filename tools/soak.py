#!/usr/bin/env python3
"""Long runs that the test-suite is too short for: (1) N synchronous steps of a BASELINE config (device-resident
cloud); (2) the asynchronous combine against the synchronous one over a long moving run (every returned array
compared).  Usage: tools/soak.py [config] [steps] [compare_steps]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "g-vom_amd")]
import numpy as np
import bench, gvom, synth
name = sys.argv[1] if len(sys.argv) > 1 else "c3"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 50000
cmp_steps = int(sys.argv[3]) if len(sys.argv) > 3 else 3000
hip = bench.Hip(); hip.set_device(0)
params, scans = synth.config_inputs(name, n_scans=8)
dev = [(hip.to_device(pc), pc.shape[0], pc.dtype, ego, tf) for (pc, ego, tf) in scans]
g = gvom.Gvom(*params)
t0 = time.perf_counter()
for k in range(steps):
    d, n, dt, ego, tf = dev[k % len(dev)]; g.process_pointcloud_device(d.value, n, dt, ego, tf); out = g.combine_maps()
elapsed = time.perf_counter() - t0                     # (the dense read-back below -- 16 bytes per voxel to pageable memory -- is not a step)
st = g.read_dense(gvom.GVOM_WHICH_FUSED)
print("%s: %d steps, %.1f us/step (first-use allocations included); fused state min %d, hit max %d, total max %d" % (
    name, steps, elapsed / steps * 1e6, st[0].min(), st[1].max(), st[2].max()), flush=True)
del g
if cmp_steps <= 0:
    sys.exit(0)
small = (0.2, 0.2, 128, 64, 4) + synth.REF_TAIL
a, b = gvom.Gvom(*small), gvom.Gvom(*small)
rng = np.random.default_rng(5); pend = None; want = None; bad = 0
for k in range(cmp_steps):
    ego = (0.05 * k, 0.02 * k, 0.0)
    pc = np.stack([rng.uniform(-12, 12, 20000) + ego[0], rng.uniform(-12, 12, 20000) + ego[1],
                   rng.normal(-0.8, 0.4, 20000)], 1).astype(np.float32)
    a.process_pointcloud(pc, ego); b.process_pointcloud(pc, ego)
    if pend is not None:
        got = pend.result()
        bad += sum(0 if np.array_equal(got[i], want[i]) else 1 for i in range(5))
    want = a.combine_maps(); pend = b.combine_maps_async()
got = pend.result(); bad += sum(0 if np.array_equal(got[i], want[i]) else 1 for i in range(5))
print("asynchronous == synchronous over %d moving steps: %d differing arrays" % (cmp_steps, bad))
sys.exit(1 if bad else 0)

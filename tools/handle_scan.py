#!/usr/bin/env python3
"""Is the 131 / 139 us bimodality of the step time a property of the process or of the handle (its
HIP stream / allocations)?  Several handles in one process, timed one after the other, twice."""
import os, sys, time, gc
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "g-vom_amd")); sys.path.insert(0, ROOT)
import gvom, synth, bench
bench.pin_to_gpu_numa(0)
params, scans = synth.config_inputs("m256", n_scans=1)
hip = bench.Hip(); hip.set_device(0)
pc, ego, tf = scans[0]
d = hip.to_device(pc)
hs = [gvom.Gvom(*params, device=0) for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 6)]
def run(g, k):
    for _ in range(k):
        g.process_pointcloud_device(d.value, pc.shape[0], pc.dtype, ego, tf); g.combine_maps()
gc.disable()
for rep in range(2):
    for j, g in enumerate(hs):
        run(g, 50)
        t = time.perf_counter(); run(g, 400); dt = (time.perf_counter() - t) / 400 * 1e6
        print("rep %d handle %d: %.1f us/step" % (rep, j, dt))

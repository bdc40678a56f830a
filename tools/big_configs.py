#!/usr/bin/env python3
"""BASELINE configs c4 (512 x 512 x 128, 1 M returns) and c5 (1024 x 1024 x 128, 4 M returns, 20 Hz
target) on ONE MI355X: step time of scan + combine with the cloud resident in HBM and as a host array."""
import os, sys, time, gc
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "g-vom_amd")); sys.path.insert(0, ROOT)
import numpy as np
import gvom, synth, bench
bench.pin_to_gpu_numa(0)
hip = bench.Hip(); hip.set_device(0)
gc.disable()
for name, xy, zs, nsens, beams, buf in (("c4", 512, 128, 4, 128, 1), ("c5", 1024, 128, 16, 128, 1), ("c5 buffer=4", 1024, 128, 16, 128, 4)):
    params = (0.2, 0.2, xy, zs, buf) + synth.REF_TAIL
    scene = synth.make_scene(2, extent=0.2 * xy / 2 * 0.9)
    egos = [(0.2 * k, -0.1 * k, 0.0) for k in range(4)]
    clouds = [np.concatenate([synth.lidar_scan(scene, beams=beams, sensor=e, yaw=2 * np.pi / 2048 * r / nsens, noise_seed=r)
                              for r in range(nsens)], 0) for e in egos]
    g = gvom.Gvom(*params, device=0)
    devs = [hip.to_device(c) for c in clouds]
    n = clouds[0].shape[0]
    def step_dev(k):
        g.process_pointcloud_device(devs[k % 4].value, n, np.float32, egos[k % 4], None); g.combine_maps()
    def step_host(k):
        g.process_pointcloud(clouds[k % 4], egos[k % 4], None); g.combine_maps()
    for label, f in (("device-resident", step_dev), ("host array", step_host)):
        for k in range(12): f(k)
        t = time.perf_counter()
        for k in range(40): f(k)
        dt = (time.perf_counter() - t) / 40
        print("%-12s %7d returns, %-15s: %8.1f us/step = %6.1f Hz, %7.1f M points/s" % (name, n, label, dt * 1e6, 1 / dt, n / dt / 1e6))
    g.set_profiling(True); step_dev(0); print("             stages us:", {k: round(v * 1e3, 1) for k, v in g.last_stage_ms().items()})
    del g

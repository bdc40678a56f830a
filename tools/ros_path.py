#!/usr/bin/env python3
"""Step time of the path an unchanged gvom_ros.py drives: process_pointcloud(float64 HOST array from
ros_numpy, ego, 4x4 transform) + combine_maps(), beside the f32 host and f32 device-resident paths."""
import os, sys, time, gc
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "g-vom_amd")); sys.path.insert(0, ROOT)
import numpy as np
import gvom, synth, bench
bench.pin_to_gpu_numa(0)
params, scans = synth.config_inputs("m256", n_scans=1)
hip = bench.Hip(); hip.set_device(0)
pc, ego, tf = scans[0]
pc64 = pc.astype(np.float64)
T = np.eye(4)
d = hip.to_device(pc)
g = gvom.Gvom(*params, device=0)
def timeit(f, k=400):
    for _ in range(50): f()
    t = time.perf_counter()
    for _ in range(k): f()
    return (time.perf_counter() - t) / k * 1e6
gc.disable()
cases = [("f32 device-resident", lambda: (g.process_pointcloud_device(d.value, pc.shape[0], pc.dtype, ego, None), g.combine_maps())),
         ("f32 host array", lambda: (g.process_pointcloud(pc, ego, None), g.combine_maps())),
         ("f64 host array (ROS path)", lambda: (g.process_pointcloud(pc64, ego, None), g.combine_maps())),
         ("f64 host array + identity transform (ROS path)", lambda: (g.process_pointcloud(pc64, ego, T), g.combine_maps()))]
for name, f in cases:
    us = timeit(f)
    print("%-50s %.1f us/step  %.0f M points/s" % (name, us, pc.shape[0] / us))
g.set_profiling(True); g.process_pointcloud(pc64, ego, T); g.combine_maps(); print({k: round(v * 1e3, 1) for k, v in g.last_stage_ms().items()})

#!/usr/bin/env python3
"""CPU oracle (oracle/gvom_oracle.c, OpenMP build) on the headline workload at 1 / 8 / 16 / 32 / 64 / 128 threads:
which thread count bench.py's cpu_baseline.value_all_cores should use on this box.  Whole steps (scan + combine)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "g-vom_amd")]
import synth
from oracle import oracle
params, scans = synth.config_inputs("m256", n_scans=4)
cores = len(os.sched_getaffinity(0))
import bench
print("logical CPUs in the affinity mask: %d; usable: %s" % (cores, bench.usable_cores()))
for f in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
    if os.path.exists(f):
        print(f, "=", open(f).read().strip())
for t in sorted(set((1, 8, 12, 16, 20, 24, 32, 64, 128, bench.usable_cores()[0]))):
    if t > cores:
        continue
    th = oracle.use_all_cores(t > 1, threads=t if t > 1 else None)
    g = oracle.OracleGvom(*params)
    g.reuse_buffers = True
    for k in range(2):
        pc, ego, tf = scans[k % 4]; g.process_pointcloud(pc, ego, tf); g.combine_maps()
    k, t0 = 0, time.perf_counter()
    while k < 3 or time.perf_counter() - t0 < 4.0:
        pc, ego, tf = scans[k % 4]; g.process_pointcloud(pc, ego, tf); g.combine_maps(); k += 1
    el = time.perf_counter() - t0
    print("%3d threads: %.3f M points/s, %.1f ms/step (%d steps)" % (th or 1, k * pc.shape[0] / el / 1e6, el / k * 1e3, k), flush=True)
oracle.use_all_cores(False)

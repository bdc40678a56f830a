#!/usr/bin/env python3
"""bench.py -- G-VOM hot path (process_pointcloud -> combine_maps) on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config m256|c2|c3|m256b8]

A "step" is one pass of the hot path over one synthetic lidar scan that is already resident
in HBM: gvom_process_pointcloud_device (transform/hit/DDA trace, encode, min-height) followed
by gvom_combine_maps (temporal fusion + column reductions + 2-D maps + D2H of the four
returned maps).  Metric: M points/s (whole job), with end-to-end map Hz beside it.

N = 1 runs the headline configuration of BASELINE.json's metric: the 256^3 voxel grid at
0.2 m with the OS1-64-shaped 131,072-point scan (BASELINE.md row "M").  N > 1 runs the
slab-sharded mapper (g-vom_amd/gvom_sharded.py): one rank per GPU, the grid partitioned into
world-anchored y-slabs, N sensors' scans per step (weak scaling: per-GPU point count fixed).

Rank 0 prints ONE JSON line with the `roofline` (dominant kernel, HIP-event timed inside the
timed region on the library's own stream) and `cpu_baseline` (the CPU oracle, a "port" of the
reference's algorithm, timed on this host's cores on a bounded sample) objects.
"""
import argparse
import ctypes
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "g-vom_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np          # noqa: E402

HBM_PEAK_GBS = 8000.0       # MI355X spec peak (MI355X_MICROARCH.md: 8.0 TB/s; 6.29 TB/s measured copy)


class Hip(object):
    """Minimal HIP runtime binding for device buffers (plumbing only)."""

    def __init__(self):
        self.rt = ctypes.CDLL("libamdhip64.so")
        self.rt.hipMalloc.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_size_t]
        self.rt.hipMemcpy.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
        self.rt.hipFree.argtypes = [ctypes.c_void_p]
        self.rt.hipSetDevice.argtypes = [ctypes.c_int]

    def chk(self, rc, what):
        if rc != 0:
            raise RuntimeError("%s failed with hipError %d" % (what, rc))

    def set_device(self, d):
        self.chk(self.rt.hipSetDevice(d), "hipSetDevice")

    def to_device(self, arr):
        p = ctypes.c_void_p()
        self.chk(self.rt.hipMalloc(ctypes.byref(p), arr.nbytes), "hipMalloc")
        self.chk(self.rt.hipMemcpy(p, arr.ctypes.data_as(ctypes.c_void_p), arr.nbytes, 1), "hipMemcpy")
        self.chk(self.rt.hipDeviceSynchronize(), "hipDeviceSynchronize")
        return p


def pin_to_gpu_numa(device):
    """Benchmark hygiene (what `numactl --cpunodebind` does): run this process on the CPUs of the
    NUMA node the GPU hangs off, so kernel launches (doorbell writes) and completion flags (GPU
    writes to host memory the host spins on) do not cross the socket interconnect (2-socket test
    hosts: 130.0-131.2 us/step pinned, 130.9-131.5 unpinned).
    Best effort: returns the CPU list used, or None (GVOM_BENCH_NO_PIN=1 disables it)."""
    if os.environ.get("GVOM_BENCH_NO_PIN") or not hasattr(os, "sched_setaffinity"):
        return None
    try:
        rt = ctypes.CDLL("libamdhip64.so")
        buf = ctypes.create_string_buffer(64)
        if rt.hipDeviceGetPCIBusId(buf, 64, int(device)) != 0:
            return None
        bdf = buf.value.decode().lower()
        with open("/sys/bus/pci/devices/%s/local_cpulist" % bdf) as f:
            spec = f.read().strip()
        cpus = set()
        for part in spec.split(","):
            if part:
                a, _, b = part.partition("-")
                cpus.update(range(int(a), int(b or a) + 1))
        cpus &= os.sched_getaffinity(0)
        if not cpus:
            return None
        os.sched_setaffinity(0, cpus)
        return spec
    except Exception:
        return None


def cpu_baseline(params, scans, budget_s=20.0):
    """Times the CPU oracle (single thread, C restatement of the reference's algorithm) on a
    bounded sample of the same workload: whole steps (one scan + one combine) until the
    budget is used, at least one."""
    from oracle import oracle
    g = oracle.OracleGvom(*params)
    pts = 0
    steps = 0
    t0 = time.perf_counter()
    while True:
        pc, ego, tf = scans[steps % len(scans)]
        g.process_pointcloud(pc, ego, tf)
        g.combine_maps()
        pts += pc.shape[0]
        steps += 1
        el = time.perf_counter() - t0
        if el > budget_s and steps >= 3:
            break
    return {"value": pts / el / 1e6, "unit": "M points/s", "cores": 1, "kind": "port",
            "sample": "%d whole steps (scan+combine) of the same workload, %.1f s, "
                      "oracle/gvom_oracle.c single thread" % (steps, el),
            "ms_per_step": el / steps * 1e3}


def pmc_traffic(kernel):
    """HBM bytes per launch of `kernel` from the committed rocprofv3 PMC passes (profiles/
    *_traffic.json, written by tools/summarize_pmc.py: separate FETCH_SIZE / WRITE_SIZE passes,
    corrected with the factors calibrated on known-byte kernels in the same session), newest
    file first.  None if no profile covers the kernel."""
    import glob
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_traffic.json")), reverse=True):
        try:
            k = json.load(open(f))["kernels"].get(kernel)
        except Exception:
            k = None
        if k and k.get("hbm_bytes_corrected"):
            return {"bytes_per_launch": k["hbm_bytes_corrected"], "source": os.path.basename(f),
                    "atomic_requests_per_launch": k.get("TCC_EA0_ATOMIC_sum")}
    return None


def atomic_ceiling(kernel, ms):
    """k_trace's memory traffic is scattered atomics: report its memory-side atomic REQUEST rate
    beside the HBM roofline.  Requests per launch come from the committed
    TCC_EA0_ATOMIC_sum pass; the ceiling is the scattered-request rate measured with
    tools/atomic_calib on the same part (profiles/r1e_atomic_calib.txt: 24.8 G requests/s when every
    request goes to a different line; requests to ONE line are served at 11.4 ns each)."""
    if kernel != "trace":
        return None
    t = pmc_traffic("k_trace") or {}
    req = t.get("atomic_requests_per_launch")
    if not req:
        return None
    return {"bound": "memory-side atomic requests", "requests_per_launch": req,
            "achieved": req / (ms * 1e-3) / 1e9, "peak": 24.8, "unit": "G requests/s",
            "frac": req / (ms * 1e-3) / 1e9 / 24.8, "source": t.get("source")}


def run_single(args):
    import gvom
    import synth
    affinity = pin_to_gpu_numa(0)
    hip = Hip()
    hip.set_device(0)
    name = args.config
    params, scans = synth.config_inputs(name, n_scans=max(1, args.poses))
    g = gvom.Gvom(*params, device=0)
    dev = [(hip.to_device(pc), pc.shape[0], pc.dtype, ego, tf) for (pc, ego, tf) in scans]
    n_pts = scans[0][0].shape[0]

    def step(k):
        d, n, dt, ego, tf = dev[k % len(dev)]
        g.process_pointcloud_device(d.value, n, dt, ego, tf)
        return g.combine_maps()

    for k in range(args.warmup):
        step(k)
    # HIP-event timing of the kernels happens INSIDE the timed region, on the library's own
    # stream, on every `sample`-th step (a sampled step is ~80 us longer: event records between the
    # kernels and a stream sync to read them)
    acc = dict.fromkeys(gvom.STAGE_NAMES, 0.0)
    sample, n_sampled = max(1, args.sample), 0
    import gc
    gc.collect(); gc.disable()            # no cyclic-GC pauses inside the timed region (buffers recycle by refcount)
    t0 = time.perf_counter()
    for k in range(args.steps):
        prof = (k % sample) == 0
        if prof:
            g.set_profiling(True)
        step(args.warmup + k)
        if prof:
            ms = g.last_stage_ms()
            g.set_profiling(False)
            n_sampled += 1
            for s in acc:
                acc[s] += ms[s]
    elapsed = time.perf_counter() - t0
    gc.enable()
    stage_ms = {s: acc[s] / n_sampled for s in acc}
    # PCIe-inclusive rate (never `value`): the same steps with the cloud handed over as a HOST
    # buffer, the reference's own calling convention (gvom.py:110 cuda.to_device)
    n_pcie = max(10, args.steps // 4)
    t1 = time.perf_counter()
    for k in range(n_pcie):
        pc, ego, tf = scans[k % len(scans)]
        g.process_pointcloud(pc, ego, tf)
        g.combine_maps()
    pcie_elapsed = time.perf_counter() - t1
    # the same steps through combine_maps_occupancy (combine + the ROS node's post-processing on the
    # GPU, 5 B/cell over PCIe instead of 20): reported beside the headline, never `value`
    n_occ = max(10, args.steps // 4)
    t2 = time.perf_counter()
    for k in range(n_occ):
        d, n, dt, ego, tf = dev[k % len(dev)]
        g.process_pointcloud_device(d.value, n, dt, ego, tf)
        g.combine_maps_occupancy()
    occ_elapsed = time.perf_counter() - t2
    # exact integer accounting for the roofline (algorithmic bytes).  AFTER the timed regions:
    # the dense read-back allocates and frees 16*V bytes, and that free shows up as one 7-15 ms step
    # shortly afterwards (tools/step_hist.py), i.e. +4-8 us on the average of 1000 steps
    stats = g.scan_stats()

    V = params[2] * params[2] * params[3]
    P = 12 if scans[0][0].dtype == np.float32 else 24
    n_in = stats["sum_hit"]
    alg = {                                                          # bytes per launch
        "trace": n_pts * P + 4 * (stats["sum_hit"] + stats["sum_total"]),
        "encode": 20 * V + n_pts * P + 4 * n_in,          # min-height pass rides in the k_encode launch
        "min_height": 0,
        "fuse": 4 * V * (min(args.poses, params[4]) + 1) + 4 * V + 4 * V,
        "map2d": 68 * params[2] * params[2],
    }
    stage_ms.pop("min_height", None)
    alg.pop("min_height", None)
    dom = max(stage_ms, key=lambda s: stage_ms[s])
    achieved = alg[dom] / (stage_ms[dom] * 1e-3) / 1e9
    out = {
        "metric": "M points/sec (process_pointcloud + combine_maps, 256^3 voxel grid); map Hz beside it",
        "value": n_pts * args.steps / elapsed / 1e6,
        "unit": "M points/s",
        "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "int32 atomics + f32 ray state + f64 compares/maps",
        "data": "synthetic",
        "config": {"workload": synth.CONFIGS[name][2], "name": name, "points_per_scan": n_pts,
                   "grid": [params[2], params[2], params[3]], "buffer_size": params[4],
                   "poses": len(scans), "input": "device-resident f32 xyz",
                   "step": "1 scan + 1 combine incl. D2H of the 4 maps",
                   "host_affinity": affinity},
        "map_hz": args.steps / elapsed,
        "value_pcie_inclusive": n_pts * n_pcie / pcie_elapsed / 1e6,
        "value_occupancy_api": n_pts * n_occ / occ_elapsed / 1e6,
        "stage_ms": stage_ms,
        "host_us": g.host_timing(),
        "sum_hit": stats["sum_hit"], "sum_total": stats["sum_total"], "cells": stats["cells"],
        "roofline": {"bound": "hbm", "kernel": "k_" + dom, "achieved": achieved, "peak": HBM_PEAK_GBS,
                     "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                     "traffic": (pmc_traffic("k_" + dom) or {}).get("bytes_per_launch"),
                     "traffic_detail": pmc_traffic("k_" + dom),
                     "algorithmic_bytes_per_launch": alg[dom], "avg_launch_ms": stage_ms[dom],
                     "secondary_ceiling": atomic_ceiling(dom, stage_ms[dom]),
                     "all_stages_GBs": {s: alg[s] / (stage_ms[s] * 1e-3) / 1e9 if stage_ms[s] > 0 else None
                                        for s in alg}},
    }
    if not args.no_cpu:
        out["cpu_baseline"] = cpu_baseline(params, scans, args.cpu_budget)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--warmup", type=int, default=100)
    ap.add_argument("--config", default="m256", choices=["c2", "c3", "m256", "m256b8"])
    ap.add_argument("--poses", type=int, default=1, help="distinct sensor poses cycled through")
    ap.add_argument("--sample", type=int, default=50,
                    help="HIP-event-time the kernels on every n-th timed step (a sampled step costs ~80 us more: "
                         "event records + a stream sync; every 8th step inflated the step average by 10 us)")
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU baseline leg")
    ap.add_argument("--cpu-budget", type=float, default=15.0)
    args = ap.parse_args()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus > 1 or world > 1 or os.environ.get("GVOM_BENCH_FORCE_SHARDED"):
        import bench_sharded
        # RCCL prints its version banner on stdout: keep stdout for the ONE JSON line (fd-level, the
        # banner comes from native code)
        sys.stdout.flush()
        saved = os.dup(1)
        os.dup2(2, 1)
        try:
            out = bench_sharded.run(args)
        finally:
            sys.stdout.flush()
            os.dup2(saved, 1)
            os.close(saved)
        if out is None:
            return
    else:
        out = run_single(args)
    print(json.dumps(out))


if __name__ == "__main__":
    main()

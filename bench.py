#!/usr/bin/env python3
"""bench.py -- G-VOM hot path (process_pointcloud -> combine_maps) on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config m256|c1|c2|c3|m256b8|c4|c5]

A "step" is one pass of the hot path over one synthetic lidar scan: process_pointcloud
(transform / hit / DDA trace, encode) followed by combine_maps (temporal fusion + column
reductions + 2-D maps + the four returned maps landing in host memory).  Metric: M points/s
(whole job), with end-to-end map Hz beside it.

`value` is measured with the cloud already resident in HBM (the driver's contract).  The same
steps through the reference's own calling conventions are reported beside it, never as `value`:
`value_host_f32` (host numpy in, gvom.py:110) and `value_ros_f64_tf` (what an unchanged
gvom_ros.py:106-109 hands over: a float64 host array + a 4x4 transform).

N = 1 runs the headline configuration of BASELINE.json's metric: the 256^3 voxel grid at 0.2 m
with the OS1-64-shaped 131,072-point scan (BASELINE.md row "M"), cycling 8 sensor poses that
move 0.2 m per scan (the window shifts).  Whatever --steps says, timed blocks of K steps are
repeated until >= 0.5 s has been timed; `ms_per_step` is the MEDIAN block.  Short runs of the
other single-GPU BASELINE configs ride along under `configs`.

N > 1 runs one map sharded over N GPUs (g-vom_amd/gvom_sharded.py): one rank per GPU, every
rank its own sensor (weak scaling: per-GPU point count fixed), RCCL over xGMI called from
libgvom_hip.so.  Started without a launcher (`python bench.py --gpus N`) this process only
spawns the N ranks -- it never touches the GPU itself -- and relays rank 0's JSON line; under
torchrun (RANK / WORLD_SIZE / MASTER_* in the environment) it is one of the ranks.

Rank 0 prints ONE JSON line with `roofline` (dominant kernel, HIP-event timed on the library's
own stream) and `cpu_baseline` (the CPU oracle, a "port" of the reference's algorithm: one thread
and all host cores, on a bounded sample of the same workload).
"""
import argparse
import ctypes
import glob
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "g-vom_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

HBM_PEAK_GBS = 8000.0       # MI355X spec peak (MI355X_MICROARCH.md: 8.0 TB/s; 6.29 TB/s measured copy)
PCIE_PEAK_GBS = 64.0        # host link, one direction: PCIe Gen5 x16 (32 GT/s x 16 lanes, 128b/130b); 42-48 GB/s is what
                            # k_map2d's write-through stores into pinned host memory reach (profiles/r4_experiments.txt)
VALU_ISSUE_RATE = 1.2288e12  # wave64 VALU instructions / s: 1024 SIMDs x 2.4 GHz / 2 cycles (MI355X_MICROARCH.md:54,473)
MIN_TIMED_S = 0.5
METRIC = "M points/sec (process_pointcloud + combine_maps, 256^3 voxel grid); map Hz beside it"


def metric_for(name, grid=None):
    """BASELINE.json's metric, naming the grid the line was measured on (m256 IS the 256^3 grid the metric is quoted on;
    the other configs are reported under their own grid, never under the 256^3 label)."""
    if name == "m256" or not grid:
        return METRIC
    return "M points/sec (process_pointcloud + combine_maps, %dx%dx%d voxel grid: BASELINE config %s); map Hz beside it" % (grid[0], grid[1], grid[2], name)
DTYPE = "int32 atomics + f32 ray state + f64 compares/maps"


class Hip(object):
    """Minimal HIP runtime binding for device buffers (plumbing only)."""

    def __init__(self):
        self.rt = ctypes.CDLL("libamdhip64.so")
        self.rt.hipMalloc.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_size_t]
        self.rt.hipMemcpy.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
        self.rt.hipFree.argtypes = [ctypes.c_void_p]
        self.rt.hipSetDevice.argtypes = [ctypes.c_int]
        self.rt.hipEventCreate.argtypes = [ctypes.POINTER(ctypes.c_void_p)]
        self.rt.hipEventRecord.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
        self.rt.hipEventSynchronize.argtypes = [ctypes.c_void_p]
        self.rt.hipEventElapsedTime.argtypes = [ctypes.POINTER(ctypes.c_float), ctypes.c_void_p, ctypes.c_void_p]

    def chk(self, rc, what):
        if rc != 0:
            raise RuntimeError("%s failed with hipError %d" % (what, rc))

    def set_device(self, d):
        self.chk(self.rt.hipSetDevice(d), "hipSetDevice")

    def device_count(self):
        n = ctypes.c_int(0)
        self.chk(self.rt.hipGetDeviceCount(ctypes.byref(n)), "hipGetDeviceCount")
        return n.value

    def event_create(self):
        e = ctypes.c_void_p()
        self.chk(self.rt.hipEventCreate(ctypes.byref(e)), "hipEventCreate")
        return e

    def event_record(self, ev, stream):
        self.chk(self.rt.hipEventRecord(ev, ctypes.c_void_p(stream)), "hipEventRecord")

    def event_elapsed_ms(self, a, b):
        """(both events must have completed: the caller has synchronised their stream)"""
        ms = ctypes.c_float(0.0)
        self.chk(self.rt.hipEventSynchronize(b), "hipEventSynchronize")
        self.chk(self.rt.hipEventElapsedTime(ctypes.byref(ms), a, b), "hipEventElapsedTime")
        return float(ms.value)

    def to_device(self, arr):
        p = ctypes.c_void_p()
        self.chk(self.rt.hipMalloc(ctypes.byref(p), arr.nbytes), "hipMalloc")
        self.chk(self.rt.hipMemcpy(p, arr.ctypes.data_as(ctypes.c_void_p), arr.nbytes, 1), "hipMemcpy")
        self.chk(self.rt.hipDeviceSynchronize(), "hipDeviceSynchronize")
        return p


def pin_to_gpu_numa(device):
    """Benchmark hygiene (what `numactl --cpunodebind` does): run this process on the CPUs of the
    NUMA node the GPU hangs off, so kernel launches (doorbell writes) and completion flags (GPU
    writes to host memory the host spins on) do not cross the socket interconnect.
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


def usable_cores(cgroup_files=("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us")):
    """(cores this process can actually run on, how that was found): the CPUs of its affinity mask, capped by the cgroup's CPU
    quota -- a container that SEES 256 logical CPUs but holds a 16-CPU share is throttled beyond 16 busy threads (measured on
    the GPU pool: every loop of the all-core oracle, even a plain fill, takes 5-70x longer at 128 threads than at 16)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    how = "affinity mask: %d" % n
    for path in cgroup_files:
        try:
            with open(path) as f:
                txt = f.read().split()
            if path.endswith("cpu.max"):
                quota, period = txt[0], float(txt[1])
            else:
                quota = txt[0]
                with open(os.path.join(os.path.dirname(path), "cpu.cfs_period_us")) as f:
                    period = float(f.read().split()[0])
            if quota not in ("max", "-1") and float(quota) > 0:
                q = max(1, int(float(quota) / period + 0.5))
                if q < n:
                    n, how = q, how + "; cgroup CPU quota: %d" % q
            break
        except (OSError, ValueError, IndexError):
            continue
    return n, how


def cpu_baseline(params, scans, budget_s=20.0):
    """Times the CPU oracle (C restatement of the reference's algorithm, oracle/gvom_oracle.c) on a
    bounded sample of the same workload -- whole steps (one scan + one combine) -- first on ONE
    thread, then its OpenMP build on ALL host cores this process may use."""
    from oracle import oracle
    cores, cores_how = usable_cores()
    out = {}
    # one thread, then the OpenMP build on ALL the cores this process may use, and on 16 threads (round 3's best) beside it.
    # (Round 3's all-core build got SLOWER beyond 16-32 threads -- rays dealt out in cloud order made every thread add into the
    # same cache lines, and numpy's one-thread np.full / fresh pages were most of a step; now a thread traces one azimuth sector
    # of the cloud, the V-sized fills run on all threads and the V-sized arrays are re-used: oracle/gvom_oracle.c, oracle.py.)
    sweep = {}
    legs = [("one", None, 0.5), ("all", cores, 0.35)] + ([("half", max(1, cores // 2), 0.15)] if cores >= 4 else [])
    for key, threads_req, share in legs:
        threads = oracle.use_all_cores(threads_req is not None, threads=threads_req)
        g = oracle.OracleGvom(*params)
        g.reuse_buffers = True
        for k in range(2 if threads_req else 0):          # (first touch of the re-used arrays)
            pc, ego, tf = scans[k % len(scans)]
            g.process_pointcloud(pc, ego, tf); g.combine_maps()
        pts = steps = 0
        t0 = time.perf_counter()
        while True:
            pc, ego, tf = scans[steps % len(scans)]
            g.process_pointcloud(pc, ego, tf)
            g.combine_maps()
            pts += pc.shape[0]
            steps += 1
            el = time.perf_counter() - t0
            if el > budget_s * share and steps >= 3:
                break
        out[key] = (pts / el / 1e6, steps, el, threads)
        if threads_req:
            sweep[str(threads)] = pts / el / 1e6
    oracle.use_all_cores(False)
    v1, s1, e1, _ = out["one"]
    vn, sn, en, tn = out["all"]
    return {"value": v1, "unit": "M points/s", "cores": 1, "kind": "port",
            "value_all_cores": vn, "cores_all": tn, "host_cores_available": cores, "host_cores_how": cores_how,
            "threads_sweep": sweep,
            "cores_all_note": "%d OpenMP threads = every core this process can run on (%s); %s M points/s by thread count in this run; "
                              "the sweep past the quota is profiles/r4_cpu_thread_sweep.txt" % (tn, cores_how, json.dumps(sweep)),
            "sample": "%d whole steps (scan+combine) of the same workload in %.1f s on one thread, %d steps in %.1f s "
                      "on %d OpenMP threads; oracle/gvom_oracle.c" % (s1, e1, sn, en, tn),
            "ms_per_step": e1 / s1 * 1e3, "ms_per_step_all_cores": en / sn * 1e3}


_CONFIG_TAGS = ("c1", "c2", "c3", "c4", "c5", "m256b8", "m256_d10", "m256_d25", "m256_d40")


def _newest_profile(pattern, kernel, config="m256"):
    """newest committed counter summary for `config`: profiles/<tag>_<config>_<kind>.json, or <tag>_<kind>.json for the
    headline workload m256"""
    kind = pattern.lstrip("*")                            # "_traffic.json" / "_sq.json"
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", pattern)), reverse=True)
    if config == "m256":
        files = [f for f in files if not any(os.path.basename(f).endswith("_%s%s" % (c, kind)) for c in _CONFIG_TAGS)]
    else:
        files = [f for f in files if os.path.basename(f).endswith("_%s%s" % (config, kind))]
    for f in files:
        try:
            ks = json.load(open(f))["kernels"]
        except Exception:
            continue
        for name, k in ks.items():
            if name.split("<")[0] == kernel:
                return k, os.path.basename(f)
    return None, None


_LIB_SHA = {}


def loaded_library_sha():
    """sha256 of the libgvom_hip.so this process loads (hipcc's output is reproducible: the sha names the sources)."""
    import hashlib
    import gvom
    path = gvom.library_path()
    if path not in _LIB_SHA:
        h = hashlib.sha256()
        with open(path, "rb") as f:
            for chunk in iter(lambda: f.read(1 << 20), b""):
                h.update(chunk)
        _LIB_SHA[path] = h.hexdigest()
    return _LIB_SHA[path]


def profile_library(source):
    """the library identity a committed counter summary was taken on ({"lib_sha256", "git_head", ...}; None: not recorded --
    summaries older than round 6)"""
    try:
        return json.load(open(os.path.join(ROOT, "profiles", source))).get("library")
    except Exception:
        return None


def pmc_traffic(kernel, config="m256"):
    """HBM bytes per launch of `kernel` from the committed rocprofv3 PMC passes (profiles/*_traffic.json:
    separate FETCH_SIZE / WRITE_SIZE passes, corrected with the factors calibrated on known-byte
    kernels in the same session), newest file first.  None if no profile covers the kernel."""
    k, src = _newest_profile("*_traffic.json", kernel, config)
    if k and k.get("hbm_bytes_corrected"):
        return {"bytes_per_launch": k["hbm_bytes_corrected"], "source": src,
                "atomic_requests_per_launch": k.get("TCC_EA0_ATOMIC_sum")}
    return None


def sq_counters(kernel, config="m256"):
    """SQ counters per launch of `kernel` from the committed passes (profiles/*_sq.json, tools/pmc_sq.sh)."""
    k, src = _newest_profile("*_sq.json", kernel, config)
    return (k, src) if k else (None, None)


def _median(xs):
    xs = sorted(xs)
    return xs[len(xs) // 2] if xs else None


def timed_blocks(step, k0, steps, min_s=MIN_TIMED_S, max_blocks=400):
    """Blocks of exactly `steps` steps until >= min_s has been timed; returns the block times (s)."""
    import gc
    gc.collect(); gc.disable()            # no cyclic-GC pauses inside the timed region (buffers recycle by refcount)
    blocks, total, k = [], 0.0, k0
    try:
        while (total < min_s or len(blocks) < 3) and len(blocks) < max_blocks:
            t0 = time.perf_counter()
            for _ in range(steps):
                step(k); k += 1
            dt = time.perf_counter() - t0
            blocks.append(dt); total += dt
    finally:
        gc.enable()
    return blocks, k


def stage_samples(g, step, k0, n=40):
    """HIP-event times of the kernels (events on the library's own stream) of n steps, outside the
    timed blocks: a sampled step is ~80 us longer (event records + a stream sync)."""
    import gvom
    acc = {s: [] for s in gvom.STAGE_NAMES}
    g.set_profiling(True)
    for k in range(n):
        step(k0 + k)
        ms = g.last_stage_ms()
        for s in acc:
            acc[s].append(ms[s])
    g.set_profiling(False)
    acc.pop("min_height", None)           # rides in k_trace / k_encode since round 2
    return {s: {"median": _median(v), "p10": sorted(v)[len(v) // 10], "p90": sorted(v)[(9 * len(v)) // 10], "samples": len(v)}
            for s, v in acc.items()}


def run_config(hip, name, steps, warmup, poses, full):
    """One BASELINE configuration on one GPU.  full: headline treatment (all calling conventions, the
    occupancy API, roofline accounting); otherwise a short device-resident run."""
    import numpy as np
    import gvom
    import synth
    params, scans = synth.config_inputs(name, n_scans=max(1, poses))
    g = gvom.Gvom(*params, device=0)
    dev = [(hip.to_device(pc), pc.shape[0], pc.dtype, ego, tf) for (pc, ego, tf) in scans]
    lengths = [pc.shape[0] for (pc, _, _) in scans]
    # (clouds of a node that drops invalid returns differ in length from scan to scan: points per step = their mean over the poses,
    # which the timed blocks cycle through)
    n_pts = lengths[0] if len(set(lengths)) == 1 else sum(lengths) / float(len(lengths))

    def step(k):
        d, n, dt, ego, tf = dev[k % len(dev)]
        g.process_pointcloud_device(d.value, n, dt, ego, tf)
        return g.combine_maps()

    for k in range(warmup):
        step(k)
    blocks, k = timed_blocks(step, warmup, steps, MIN_TIMED_S if full else 0.1)
    med = _median(blocks)
    out = {"workload": synth.CONFIGS[name][2], "points_per_scan": n_pts, "points_per_scan_range": [min(lengths), max(lengths)],
           "grid": [params[2], params[2], params[3]],
           "buffer_size": params[4], "poses": len(scans), "steps": steps, "blocks": len(blocks),
           "ms_per_step": med / steps * 1e3, "ms_per_step_min": min(blocks) / steps * 1e3,
           "ms_per_step_max": max(blocks) / steps * 1e3, "value": n_pts * steps / med / 1e6, "map_hz": steps / med}
    stages = stage_samples(g, step, k, 40 if full else 20)
    out["stage_ms"] = {s: v["median"] for s, v in stages.items()}
    if not full:
        out["fast_path"] = {"eager_adopted": g.get_tuning("eager_adopted"), "dirsort": g.get_tuning("dirsort"), "interleave": g.get_tuning("interleave")}
        del g
        return out, None
    out["stage_ms_spread"] = stages

    # the same steps through the reference's calling conventions (never `value`)
    def step_host(k):
        pc, ego, tf = scans[k % len(scans)]
        g.process_pointcloud(pc, ego, tf)
        return g.combine_maps()

    b2, k = timed_blocks(step_host, k, steps, 0.2)
    out["value_host_f32"] = n_pts * steps / _median(b2) / 1e6
    scans64 = [(pc.astype(np.float64), ego, np.eye(4)) for (pc, ego, tf) in scans]   # gvom_ros.py:106-109

    def step_ros(k):
        pc, ego, tf = scans64[k % len(scans64)]
        g.process_pointcloud(pc, ego, tf)
        return g.combine_maps()

    b3, k = timed_blocks(step_ros, k, steps, 0.2)
    out["value_ros_f64_tf"] = n_pts * steps / _median(b3) / 1e6
    out.update(ros_two_threads(g, scans64, n_pts))
    out.update(ros_two_threads(g, scans64, n_pts, paced=True))

    # combine_maps_occupancy (combine + the ROS node's post-processing on the GPU, 5 B/cell over PCIe)
    def step_occ(k):
        d, n, dt, ego, tf = dev[k % len(dev)]
        g.process_pointcloud_device(d.value, n, dt, ego, tf)
        return g.combine_maps_occupancy()

    b4, k = timed_blocks(step_occ, k, steps, 0.2)
    out["value_occupancy_api"] = n_pts * steps / _median(b4) / 1e6

    # combine_maps_async (an extension of the reference's API): scan k+1 is handed over while the maps of
    # combine k are still being stored to host memory; every step still delivers its four maps
    pend = [None]

    def step_async(k):
        d, n, dt, ego, tf = dev[k % len(dev)]
        g.process_pointcloud_device(d.value, n, dt, ego, tf)
        maps = pend[0].result() if pend[0] is not None else None
        pend[0] = g.combine_maps_async()
        return maps

    b5, k = timed_blocks(step_async, k, steps, 0.3)
    pend[0].result()
    out["value_async_combine"] = n_pts * steps / _median(b5) / 1e6
    out["ms_per_step_async_combine"] = _median(b5) / steps * 1e3

    def step_async_occ(k):                             # the same with the int8 occupancy grids (5 B/cell over PCIe)
        d, n, dt, ego, tf = dev[k % len(dev)]
        g.process_pointcloud_device(d.value, n, dt, ego, tf)
        maps = pend[0].result() if pend[0] is not None else None
        pend[0] = g.combine_maps_occupancy_async()
        return maps

    pend[0] = None
    b6, k = timed_blocks(step_async_occ, k, steps, 0.2)
    pend[0].result()
    out["value_async_occupancy_api"] = n_pts * steps / _median(b6) / 1e6
    if os.environ.get("GVOM_HOST_TIMING"):                 # (the library only keeps these times when asked to)
        out["host_us"] = g.host_timing()
    # which path the timed steps took (results never depend on it; speed does): eager fusion adopted by the combines, the
    # cloud traced in its own order (dirsort 0) with how many sub-clouds interleaved
    out["fast_path"] = {"eager_adopted": g.get_tuning("eager_adopted"), "eager_dropped": g.get_tuning("eager_dropped"),
                        "dirsort": g.get_tuning("dirsort"), "interleave": g.get_tuning("interleave")}
    # exact integer accounting for the roofline (algorithmic bytes), AFTER the timed regions:
    # the dense read-back allocates and frees 16*V bytes
    stats = g.scan_stats()
    V = params[2] * params[2] * params[3]
    P = 12 if scans[0][0].dtype == np.float32 else 24
    filled = min(warmup + steps, params[4])
    # (4 * N_in: one f32 min per in-grid return, SURVEY 8d -- N_in = sum_hit: every in-grid return adds 1 to its voxel's hit)
    alg = {"trace": n_pts * P + 4 * (stats["sum_hit"] + stats["sum_total"]) + 4 * stats["sum_hit"],
           "encode": 20 * V + 4 * stats["sum_hit"],
           "fuse": 4 * V * (filled + 1) + 4 * V + 4 * V,
           "map2d": 68 * params[2] * params[2]}
    out.update({"sum_hit": stats["sum_hit"], "sum_total": stats["sum_total"], "cells": stats["cells"]})
    del g
    return out, (alg, stages, params, scans)


def ros_two_threads(g, scans64, n_pts, min_s=0.6, paced=False):
    """The node's real calling pattern (gvom_ros.py:61-62, 82-115): a lidar thread hands float64 clouds + a 4x4 transform
    to process_pointcloud while a timer thread calls combine_maps on the same mapper.  paced = False: both as fast as
    they can (the timer then combines about twice per scan); paced = True: the timer combines once per new scan, so
    the upload and the trace of scan k + 1 overlap the maps of combine k (ctypes drops the GIL around the library calls)."""
    import threading
    stop = threading.Event()
    n = {"scans": 0, "maps": 0}
    err = []
    sfx = "_paced" if paced else ""

    def lidar():
        k = 0
        try:
            while not stop.is_set():
                pc, ego, tf = scans64[k % len(scans64)]
                g.process_pointcloud(pc, ego, tf)
                k += 1
                n["scans"] = k
        except Exception as e:                          # pragma: no cover
            err.append(e)

    def timer():
        seen = 0
        try:
            while not stop.is_set():
                if paced:
                    if n["scans"] == seen:
                        time.sleep(0)                   # (yield: the lidar thread holds the GIL only between its calls)
                        continue
                    seen = n["scans"]
                if g.combine_maps() is not None:
                    n["maps"] += 1
        except Exception as e:                          # pragma: no cover
            err.append(e)

    ts = [threading.Thread(target=lidar), threading.Thread(target=timer)]
    for t in ts:
        t.start()
    time.sleep(0.15)                                    # warm-up
    s0, m0, t0 = n["scans"], n["maps"], time.perf_counter()
    time.sleep(min_s)
    s1, m1, t1 = n["scans"], n["maps"], time.perf_counter()
    stop.set()
    for t in ts:
        t.join(30)
    if err:
        raise err[0]
    return {"value_ros_two_threads" + sfx: (s1 - s0) * n_pts / (t1 - t0) / 1e6,
            "scans_per_s_ros_two_threads" + sfx: (s1 - s0) / (t1 - t0),
            "maps_per_s_ros_two_threads" + sfx: (m1 - m0) / (t1 - t0)}


def _wait_until(t):
    """sleep up to ~0.3 ms before t, spin the rest (time.sleep overshoots by 50-100 us)"""
    while True:
        now = time.perf_counter()
        if now >= t:
            return now
        if t - now > 4e-4:
            time.sleep(t - now - 3e-4)


def _pct(xs, q):
    ys = sorted(xs)
    return ys[min(len(ys) - 1, int(q * len(ys)))]


def paced_stream(tick, hz, ticks, t0=None, warm=3):
    """BASELINE.json config 5 as it is quoted -- "multi-sensor stream at 20 Hz, sustained throughput" (gvom_ros.py:39, 61-62:
    the node's timer; launch/gvom_node.launch:24) -- for any config: an OFFERED LOAD.  Cloud k arrives at t0 + k / hz whether
    or not the mapper is ready (a late start is queueing and counts as latency); tick(k) hands a HOST-resident cloud over
    (the upload is inside) and returns with the maps in host memory.  Per-tick latency = maps ready - scheduled arrival."""
    period = 1.0 / hz
    for k in range(warm):
        tick(k)
    import gc
    gc.collect(); gc.disable()
    if t0 is None:
        t0 = time.perf_counter() + 0.02
    lat, service, late, e = [], [], 0, 0.0
    for k in range(ticks):
        tk = t0 + k * period
        if e > tk:
            late += 1                                   # the previous tick was still running when this cloud arrived
        s = _wait_until(tk)
        tick(warm + k)
        e = time.perf_counter()
        lat.append(e - tk); service.append(e - s)
    elapsed = max(time.perf_counter() - t0, ticks * period)
    gc.enable()
    busy = sum(service)
    return {"offered_hz": hz, "ticks": ticks, "period_ms": period * 1e3,
            "achieved_hz": ticks / elapsed,
            "latency_ms": {"p50": _pct(lat, 0.5) * 1e3, "p95": _pct(lat, 0.95) * 1e3, "p99": _pct(lat, 0.99) * 1e3,
                           "max": max(lat) * 1e3, "mean": sum(lat) / len(lat) * 1e3},
            "service_ms": {"p50": _pct(service, 0.5) * 1e3, "p95": _pct(service, 0.95) * 1e3, "max": max(service) * 1e3},
            "deadline_misses": sum(1 for v in lat if v > period), "late_starts": late,
            "idle_frac": max(0.0, 1.0 - busy / elapsed), "sustainable_hz": 1.0 / _pct(service, 0.5),
            "what": "offered load: host-resident float32 cloud k handed over at t0 + k / hz (upload inside), "
                    "process_pointcloud + combine_maps per tick; latency = maps in host memory - scheduled arrival; "
                    "deadline = one period; idle_frac = 1 - time inside the two calls / elapsed (what is left for other work "
                    "on this GPU and host thread); sustainable_hz = 1 / median service time"}


def stream_single(name, hz, ticks, poses):
    """The paced stream on one GPU: one thread (the two calls back to back per tick) and the node's TWO-thread pattern (a
    lidar thread hands clouds over at hz, a timer thread combines after each: gvom_ros.py:44-51, 61-62)."""
    import threading
    import gvom
    import synth
    params, scans = synth.config_inputs(name, n_scans=max(1, min(poses, 4)))
    g = gvom.Gvom(*params, device=0)
    n_pts = scans[0][0].shape[0]

    def tick(k):
        pc, ego, tf = scans[k % len(scans)]
        g.process_pointcloud(pc, ego, tf)
        return g.combine_maps()

    out = paced_stream(tick, hz, ticks)
    out.update(points_per_tick=n_pts, offered_M_points_s=n_pts * hz / 1e6, sustained_M_points_s=n_pts * out["achieved_hz"] / 1e6,
               interleave=g.get_tuning("interleave"))
    # two threads: the lidar callback uploads + traces cloud k + 1 while the timer callback's combine k stores its maps
    period = 1.0 / hz
    lat, err = [], []
    ready = threading.Semaphore(0)
    arrivals = {}
    t0 = time.perf_counter() + 0.05

    def lidar():
        try:
            for k in range(ticks):
                tk = t0 + k * period
                _wait_until(tk)
                pc, ego, tf = scans[k % len(scans)]
                g.process_pointcloud(pc, ego, tf)
                arrivals[k] = tk
                ready.release()
        except Exception as e:                          # pragma: no cover
            err.append(e); ready.release()

    def timer():
        try:
            for k in range(ticks):
                ready.acquire()
                if err:
                    return
                g.combine_maps()
                lat.append(time.perf_counter() - arrivals[k])
        except Exception as e:                          # pragma: no cover
            err.append(e)

    ts = [threading.Thread(target=lidar, daemon=True), threading.Thread(target=timer, daemon=True)]
    for t in ts:
        t.start()
    for t in ts:
        t.join(ticks * period + 120)
    if err:
        raise err[0]
    if any(t.is_alive() for t in ts) or len(lat) != ticks:
        # (a stalled thread still holds `g`: report it instead of computing percentiles of a partial list)
        raise RuntimeError("paced two-thread stream stalled: %d of %d ticks completed" % (len(lat), ticks))
    out["two_threads"] = {"latency_ms": {"p50": _pct(lat, 0.5) * 1e3, "p95": _pct(lat, 0.95) * 1e3, "p99": _pct(lat, 0.99) * 1e3,
                                         "max": max(lat) * 1e3},
                          "deadline_misses": sum(1 for v in lat if v > period),
                          "what": "lidar thread: process_pointcloud at the offered rate; timer thread: combine_maps after every scan"}
    del g
    return out



def step_roofline(alg, res, config):
    """The whole step against the HBM peak, both ways SURVEY 8(d) asks for: the ALGORITHMIC bytes of the reference's data
    model, (B_scan + B_comb) / (t_scan + t_comb) -- V-sized clears and per-source V-sized reads included, which this
    implementation does not perform (tile epochs) -- and the bytes the counters measured, per step."""
    t = res["ms_per_step"] * 1e-3
    a = sum(alg.values())
    meas = None
    # (the fusion kernel of the config: k_fuse1 for one-slot rings, k_fuse4 otherwise -- whichever the committed passes hold)
    import synth
    one_slot = config in synth.CONFIGS and synth.CONFIGS[config][0][4] == 1
    fuse = (pmc_traffic("k_fuse1", config) if one_slot else None) or pmc_traffic("k_fuse4", config)
    parts = [pmc_traffic(k, config) for k in ("k_trace", "k_encode")] + [fuse, pmc_traffic("k_map2d", config)]
    if one_slot and res.get("stage_ms", {}).get("fuse") == 0.0:
        # eager fusion: ONE kernel (k_encfuse) encodes the slot and fuses it, right behind k_trace
        parts = [pmc_traffic(k, config) for k in ("k_trace", "k_encfuse", "k_map2d")]
    if all(parts):
        meas = sum(p["bytes_per_launch"] for p in parts)
    return {"algorithmic_bytes": a, "frac_algorithmic": a / t / 1e9 / HBM_PEAK_GBS,
            "measured_bytes": meas, "frac_measured": (meas / t / 1e9 / HBM_PEAK_GBS) if meas else None,
            "note": "per step (1 scan + 1 combine) over ms_per_step; algorithmic = N*P + 4*(sum_hit + sum_total) + 4*N_in + 20*V "
                    "+ 4*V*(S+L) + 8*V + 68*xy^2 (SURVEY 8d); measured = PMC HBM bytes of the four kernels (profiles/)"}


def roofline_of(alg, stages, profiled="m256", xy=None, lib_sha=None):
    """profiled: the config whose committed counter passes (profiles/) apply -- a config without passes of its own gets
    null counter-derived fields rather than another workload's numbers (False / None: no counters at all).
    xy: the grid's width (adds the host-link ceiling of k_map2d); lib_sha: sha256 of the library this run loaded (the line then
    says whether the committed counters were taken on the same one)."""
    config = profiled if isinstance(profiled, str) else ("m256" if profiled else None)
    kern = {"trace": "k_trace", "encode": "k_encode", "fuse": "k_fuse4", "map2d": "k_map2d"}
    try:
        import synth
        if config and synth.CONFIGS[config][0][4] == 1 and pmc_traffic("k_fuse1", config):
            kern["fuse"] = "k_fuse1"                     # one-slot rings (m256, c1, c2)
    except (ImportError, KeyError):
        pass
    ms = {s: v["median"] for s, v in stages.items()}
    if config and ms.get("fuse") == 0.0 and ms.get("encode"):
        # one-slot rings with eager fusion: the "encode" stage is k_encfuse = the slot's encoding AND its fusion (the combine
        # launches k_map2d only): its algorithmic bytes are both stages'
        kern["encode"] = "k_encfuse"
        alg = dict(alg, encode=alg["encode"] + alg["fuse"], fuse=0)
    dom = max(ms, key=lambda s: ms[s])
    achieved = alg[dom] / (ms[dom] * 1e-3) / 1e9
    traf = pmc_traffic(kern[dom], config) if config else None
    sq, sq_src = sq_counters(kern[dom], config) if config else (None, None)
    valu = None
    if sq and sq.get("SQ_INSTS_VALU"):
        bound_us = sq["SQ_INSTS_VALU"] / VALU_ISSUE_RATE * 1e6
        valu = {"insts_per_launch": sq["SQ_INSTS_VALU"], "issue_rate_per_s": VALU_ISSUE_RATE,
                "bound_us": bound_us, "frac": bound_us / (ms[dom] * 1e3),
                "measured_quad_cycles_per_inst": (sq.get("SQ_ACTIVE_INST_VALU") or 0) / sq["SQ_INSTS_VALU"],
                # the same ceiling at the rate this kernel's instruction mix is MEASURED to issue at (SQ_ACTIVE_INST_VALU quad-cycles per
                # instruction; 1.0 = 4 cycles, twice the nominal 2): what the kernel is actually up against
                "frac_at_measured_issue_rate": (bound_us / (ms[dom] * 1e3)) * 2.0 * ((sq.get("SQ_ACTIVE_INST_VALU") or 0) / sq["SQ_INSTS_VALU"]),
                "note": "frac = the kernel's VALU wave-instructions / 1.2288e12 per s / its measured time; the SQ "
                        "counters of this kernel show one quad-cycle (4 cycles) of SQ_ACTIVE_INST_VALU per "
                        "instruction, i.e. half that rate is what the integer / compare mix sustains", "source": sq_src}
    # measured (PMC) HBM traffic per stage over measured time -- NOT the algorithmic count, which assumes
    # V-sized streams that the tile tags no longer perform
    stage_gbs = {}
    for s, kname in kern.items():
        t = pmc_traffic(kname, config) if config else None
        if t and ms.get(s):
            stage_gbs[s] = t["bytes_per_launch"] / (ms[s] * 1e-3) / 1e9
    req = (traf or {}).get("atomic_requests_per_launch")
    plib = profile_library(traf["source"]) if traf else None
    traffic_sha = (plib or {}).get("lib_sha256")
    pcie = None
    if xy and ms.get("map2d"):
        # k_map2d IS host-link time: its four returned maps (3 x int32 + 1 x f64 = 20 B per cell) are stored straight into
        # pinned host memory; its HBM traffic is a few MB (0.03 of the HBM peak)
        b = 20.0 * xy * xy
        pcie = {"bound": "pcie", "kernel": "k_map2d", "bytes_to_host": b, "stage_ms": ms["map2d"],
                "achieved": b / (ms["map2d"] * 1e-3) / 1e9, "peak": PCIE_PEAK_GBS, "unit": "GB/s",
                "frac": b / (ms["map2d"] * 1e-3) / 1e9 / PCIE_PEAK_GBS,
                "note": "20 B per cell over the host link / the stage's HIP-event time / PCIe Gen5 x16's 64 GB/s"}
    return {"bound": "hbm", "kernel": kern[dom], "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": achieved / HBM_PEAK_GBS, "traffic": (traf or {}).get("bytes_per_launch"),
            "traffic_source": ("profiles/%s (committed PMC passes of this command; NOT measured by this run)" % traf["source"]) if traf else None,
            "lib_sha": lib_sha, "traffic_lib_sha": traffic_sha, "traffic_git_head": (plib or {}).get("git_head"),
            # the committed counters belong to ANOTHER library than the one measured here (or do not say which)
            "traffic_stale": (traffic_sha != lib_sha) if (traf and lib_sha) else None,
            "pcie": pcie,
            "traffic_detail": traf,
            "algorithmic_bytes_per_launch": alg[dom], "avg_launch_ms": ms[dom],
            "launch_ms_spread": {k: stages[dom][k] for k in ("p10", "p90", "samples")},
            "valu": valu,
            "secondary_ceiling": ({"bound": "memory-side atomic requests", "requests_per_launch": req,
                                   "achieved": req / (ms[dom] * 1e-3) / 1e9, "peak": 24.8, "unit": "G requests/s",
                                   "frac": req / (ms[dom] * 1e-3) / 1e9 / 24.8} if req and dom == "trace" else None),
            "measured_hbm_GBs_per_stage": stage_gbs}


def run_with_statistics(hip, name, steps, warmup, poses):
    """The same steps with the reference's per-voxel statistics switched on (gvom.py:1172-1299, 858-909: mean / covariance
    of every occupied voxel's 27-neighbourhood, merged over the ring -- what process_pointcloud and combine_maps ALSO do
    in the reference, feeding only make_debug_voxel_map).  Device-resident cloud, synchronous API."""
    import gvom
    import synth
    params, scans = synth.config_inputs(name, n_scans=max(1, poses))
    g = gvom.Gvom(*params, device=0, voxel_statistics=True)
    dev = [(hip.to_device(pc), pc.shape[0], pc.dtype, ego, tf) for (pc, ego, tf) in scans]

    def step(k):
        d, n, dt, ego, tf = dev[k % len(dev)]
        g.process_pointcloud_device(d.value, n, dt, ego, tf)
        return g.combine_maps()

    for k in range(warmup):
        step(k)
    blocks, k = timed_blocks(step, warmup, steps, 0.3)
    med = _median(blocks)
    cloud = g.make_debug_voxel_map()
    out = {"value": scans[0][0].shape[0] * steps / med / 1e6, "ms_per_step": med / steps * 1e3,
           "debug_voxel_rows": int(cloud.shape[0]) if cloud is not None else None}
    del g
    return out


def run_single(args):
    import synth
    affinity = pin_to_gpu_numa(0)
    hip = Hip()
    hip.set_device(0)
    name = args.config
    big = name in ("c4", "c5")
    poses = 1 if name == "c1" else (min(args.poses, 4) if big else args.poses)
    steps = min(args.steps, 40) if big else args.steps            # a c5 step is ~3 ms; 4 M-point clouds take a while to generate
    res, extra = run_config(hip, name, steps, min(args.warmup, 10) if big else args.warmup, poses, True)
    args.steps = steps
    alg, stages, params, scans = extra
    out = {
        "metric": metric_for(name, res["grid"]), "value": res["value"], "unit": "M points/s",
        "n_gpus": 1, "steps": args.steps, "warmup": args.warmup, "ms_per_step": res["ms_per_step"],
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": DTYPE, "data": "synthetic",
        "config": {"workload": synth.CONFIGS[name][2], "name": name, "points_per_scan": res["points_per_scan"],
                   "grid": res["grid"], "buffer_size": res["buffer_size"], "poses": res["poses"],
                   "pose_motion": "0.2 m per scan in +x (the window shifts every scan)",
                   "input": "device-resident f32 xyz", "step": "1 scan + 1 combine incl. the 4 maps in host memory",
                   "timing": "median of %d blocks of %d steps (>= %.1f s timed)" % (res["blocks"], args.steps, MIN_TIMED_S),
                   "host_affinity": affinity},
        "map_hz": res["map_hz"],
        "ms_per_step_min": res["ms_per_step_min"], "ms_per_step_max": res["ms_per_step_max"], "blocks": res["blocks"],
        "value_host_f32": res["value_host_f32"], "value_ros_f64_tf": res["value_ros_f64_tf"],
        "value_occupancy_api": res["value_occupancy_api"],
        "value_async_combine": res["value_async_combine"], "ms_per_step_async_combine": res["ms_per_step_async_combine"],
        "value_async_occupancy_api": res["value_async_occupancy_api"],
        "value_semantics": "value: cloud resident in HBM (driver contract); value_host_f32: host numpy in (gvom.py:110); "
                           "value_ros_f64_tf: float64 host array + 4x4 transform, the unchanged gvom_ros.py:106-109 path, one thread; "
                           "value_ros_two_threads: the same inputs from a lidar thread while a timer thread calls combine_maps "
                           "(the node's two callbacks, gvom_ros.py:61-62, 113-115), both free-running (_paced: one combine per new scan); "
                           "value_async_combine: combine_maps_async() (extension), the next scan traced while the maps "
                           "of the pending combine are stored to host memory -- same maps, one step later",
        "stage_ms": res["stage_ms"], "fast_path": res["fast_path"],
        "sum_hit": res["sum_hit"], "sum_total": res["sum_total"], "cells": res["cells"],
        "roofline": roofline_of(alg, stages, profiled=name, xy=params[2], lib_sha=loaded_library_sha()),
    }
    if "host_us" in res:
        out["host_us"] = res["host_us"]
    out["roofline"]["step"] = step_roofline(alg, res, name)
    for key in ("value_ros_two_threads", "scans_per_s_ros_two_threads", "maps_per_s_ros_two_threads"):
        out[key] = res[key]
        out[key + "_paced"] = res[key + "_paced"]
    if not args.no_extra:
        st = run_with_statistics(hip, name, min(steps, 200), min(args.warmup, 40), poses)
        out["value_with_statistics"] = st["value"]
        out["statistics"] = dict(st, extra_ms_per_step=st["ms_per_step"] - res["ms_per_step"],
                                 note="voxel_statistics=True: the per-voxel mean / covariance path of the reference at EVERY step (the class "
                                      "default computes it on demand: for as long as make_debug_voxel_map is being called, as the "
                                      "unchanged node does every tick; the headline steps are those of a caller that does not ask)")
    if not args.no_extra and name == "m256":
        out["configs"] = {}
        # (c4 / c5: BASELINE's multi-GPU configs on this ONE GPU, short device-resident runs -- their paced 20 Hz streams and
        # sharded forms are `--config c4|c5 [--gpus N] --offered-hz 20`)
        # (m256_d*: what a real node delivers -- non-uniform beam elevations, 10 / 25 / 40 % of the returns dropped before the call,
        # a different length every scan: the layout probe's verdicts and the trace's speed on such clouds)
        for other, poses, nsteps, nwarm in (("c2", 8, 100, 30), ("c3", 8, 100, 30), ("m256b8", 8, 100, 30), ("m256_d10", 8, 96, 32),
                                            ("m256_d25", 8, 96, 32), ("m256_d40", 8, 96, 32), ("c4", 2, 40, 10), ("c5", 1, 20, 6)):
            if getattr(args, "no_big", False) and other in ("c4", "c5"):
                continue
            r, _ = run_config(hip, other, nsteps, nwarm, poses, False)
            out["configs"][other] = r
    if args.offered_hz > 0:
        out["stream"] = dict(stream_single(name, args.offered_hz, args.ticks, poses), config=name)
    elif not args.no_extra and name == "m256" and not getattr(args, "no_big", False):
        # BASELINE.json config 5 as it is quoted -- "4 M-pt multi-sensor stream at 20 Hz, sustained throughput" -- on this ONE GPU:
        # 200 ticks (10 s) of host-resident 4,194,304-point clouds handed over at 20 Hz, upload inside
        out["stream"] = dict(stream_single("c5", 20.0, 200, 1), config="c5",
                             grid=list(synth.CONFIGS["c5"][0][2:4]), workload=synth.CONFIGS["c5"][2])
    if not args.no_cpu:
        out["cpu_baseline"] = cpu_baseline(params, scans, args.cpu_budget)
    return out


def spawn_ranks(args):
    """`python bench.py --gpus N` without a launcher: this process starts the N ranks (one per GPU) and
    relays rank 0's JSON line.  It makes NO HIP call itself."""
    import bench_sharded
    try:
        bench_sharded.workload(args.config, args.gpus)     # an unknown config / rank count is refused before anything starts
    except ValueError as e:
        sys.stderr.write("bench.py: %s\n" % e)
        sys.exit(2)
    # the device count comes from a CHILD process (this one never makes a HIP call)
    probe = subprocess.run([sys.executable, "-c", "import ctypes; rt = ctypes.CDLL('libamdhip64.so'); n = ctypes.c_int(0); "
                            "rc = rt.hipGetDeviceCount(ctypes.byref(n)); print(n.value if rc == 0 else 0)"],
                           capture_output=True, text=True)
    try:
        ndev = int(probe.stdout.strip().splitlines()[-1])
    except Exception:
        ndev = 0
    if ndev < (1 if args.share_device else args.gpus):
        sys.stderr.write("bench.py: --gpus %d, but %d HIP device(s) are visible\n" % (args.gpus, ndev))
        sys.exit(2)
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), GVOM_JOB_NONCE=str(os.getpid()), HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    out0, _ = procs[0].communicate()
    rcs = [procs[0].returncode] + [p.wait() for p in procs[1:]]
    lines = [ln for ln in out0.decode().splitlines() if ln.startswith("{")]
    if lines:
        print(lines[-1])                                   # (also when the verdict inside it is "not equal": exit code 3)
    if any(rcs):
        sys.stderr.write("bench.py: rank exit codes %s\n" % rcs)
        sys.exit(3 if rcs[0] == 3 else 1)
    if not lines:
        sys.stderr.write("bench.py: rank 0 printed no JSON line\n")
        sys.exit(1)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=500)
    ap.add_argument("--warmup", type=int, default=100)
    ap.add_argument("--config", default="m256", choices=["c1", "c2", "c3", "m256", "m256b8", "c4", "c5", "m256_d10", "m256_d25", "m256_d40"])
    ap.add_argument("--poses", type=int, default=8, help="distinct sensor poses cycled through (0.2 m apart)")
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU baseline leg")
    ap.add_argument("--no-big", action="store_true", help="skip the c4 / c5 ride-along runs of the default line")
    ap.add_argument("--no-extra", action="store_true", help="skip the short runs of the other configs")
    ap.add_argument("--cpu-budget", type=float, default=20.0)
    ap.add_argument("--offered-hz", type=float, default=0.0,
                    help="adds a paced-stream phase (BASELINE config 5: \"stream at 20 Hz\"): host-resident clouds handed over at "
                         "this rate, per-tick latency distribution and deadline misses in the line's \"stream\" object")
    ap.add_argument("--ticks", type=int, default=200, help="ticks of the paced stream")
    ap.add_argument("--transport", default="auto", choices=["auto", "rccl", "peer"],
                    help="N > 1: device data between the ranks over RCCL, by peer copies (exported regions pulled with "
                         "hipMemcpyAsync), or RCCL with peer copies as the fallback when RCCL cannot initialise (default)")
    ap.add_argument("--share-device", action="store_true",
                    help="N > 1 REHEARSAL on a box with fewer GPUs: every rank uses device 0 (peer copies; RCCL refuses it). "
                         "The line says so; it is not a scaling measurement")
    args = ap.parse_args()
    world = int(os.environ.get("WORLD_SIZE", "0"))
    if args.gpus > 1 and world == 0:
        return spawn_ranks(args)
    if args.gpus > 1 or world > 1 or os.environ.get("GVOM_BENCH_FORCE_SHARDED"):    # (the last: one rank through the sharded path)
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
            if bench_sharded.FELL_BACK:                      # (see below: no destructors under a thread that may sit inside RCCL)
                sys.stdout.flush(); sys.stderr.flush()
                os._exit(0)
            return
    else:
        out = run_single(args)
    print(json.dumps(out))
    bad = out.get("sharded_equals_unsharded") is False     # the sharded map differed from the unsharded one: the numbers above belong to wrong maps
    if "RCCL could not initialise" in str(out.get("transport", "")):
        # a helper thread may still sit inside ncclCommInitRank (gvom_comm_create2, AUTO): leave without running RCCL's destructors under it
        sys.stdout.flush(); sys.stderr.flush()
        os._exit(3 if bad else 0)
    if bad:
        sys.stdout.flush()
        sys.exit(3)


if __name__ == "__main__":
    main()

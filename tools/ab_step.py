#!/usr/bin/env python3
"""In-process A/B of k_trace settings: alternates GVOM_TRACE_* environment settings from step to step
(the library reads them at every scan) and reports the HIP-event time of every stage per setting, so
clock/box drift hits all settings alike.
Usage: tools/ab_step.py [--config m256] [--steps 400] "GVOM_TRACE_DEBUG=0" "GVOM_TRACE_DEBUG=16384" ..."""
import os, sys, ctypes
os.environ.setdefault("GVOM_ENV_DYNAMIC", "1")     # the library re-reads its switches at every call only if some GVOM_ variable exists
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "g-vom_amd")); sys.path.insert(0, ROOT)
import gvom, synth
from bench import Hip
args = sys.argv[1:]
config, steps, shard = "m256", 400, None
while args and args[0].startswith("--"):
    if args[0] == "--config": config = args[1]
    if args[0] == "--steps": steps = int(args[1])
    if args[0] == "--shard": shard = tuple(int(v) for v in args[1].split("/"))     # rank/world: one slab, whole weak-scaling cloud
    args = args[2:]
settings = [dict(kv.split("=") for kv in a.split()) for a in args]
keys = sorted({k for s in settings for k in s})
params, scans = synth.config_inputs(config, n_scans=1)
hip = Hip(); hip.set_device(0)
pc, ego, tf = scans[0]
if shard:
    params_, beams, _ = synth.CONFIGS[config]
    scene = synth.make_scene(2)
    pc = np.concatenate([synth.lidar_scan(scene, beams=beams, yaw=2 * np.pi / 2048 * r / shard[1], noise_seed=r)
                         for r in range(shard[1])], 0)
    g = gvom.Gvom(*params, device=0, _shard=shard)
else:
    g = gvom.Gvom(*params, device=0)
d = hip.to_device(pc)
def apply(s):
    for k in keys:
        if k in s: os.environ[k] = s[k]
        else: os.environ.pop(k, None)
def combine():
    if shard:
        g._lib.gvom_combine_fuse(g._h, None); g._lib.gvom_sync(g._h)
    else:
        g.combine_maps()
for _ in range(30):
    g.process_pointcloud_device(d.value, pc.shape[0], pc.dtype, ego, tf); combine()
g.set_profiling(True)
acc = [[] for _ in settings]
stage = os.environ.get("AB_STAGE", "trace")
for k in range(steps):
    i = k % len(settings)
    apply(settings[i])
    g.process_pointcloud_device(d.value, pc.shape[0], pc.dtype, ego, tf); combine()
    acc[i].append(g.last_stage_ms()[stage] * 1e3)
for a, v in zip(args, acc):
    v = np.array(v)
    print("%-50s %s us: median %.2f mean %.2f p10 %.2f p90 %.2f (n=%d)" % (a, stage, np.median(v), v.mean(), np.percentile(v, 10), np.percentile(v, 90), len(v)))

#!/usr/bin/env python3
"""Step time of the three ways to call the combine (device-resident cloud): synchronous combine_maps();
combine_maps_async().result() at once (second stream, no overlap); the pipelined use (next scan handed
over before result()).  Usage: tools/async_modes.py [config] [steps]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "g-vom_amd")]
import bench, gvom, synth
knobs = [a for a in sys.argv[1:] if "=" in a]                    # result-neutral tuning: name=value (gvom_set_tuning)
sys.argv = [a for a in sys.argv if "=" not in a]
name = sys.argv[1] if len(sys.argv) > 1 else "m256"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
hip = bench.Hip(); hip.set_device(0)
params, scans = synth.config_inputs(name, n_scans=8)
dev = [(hip.to_device(pc), pc.shape[0], pc.dtype, ego, tf) for (pc, ego, tf) in scans]
g = gvom.Gvom(*params)
for kv in knobs:
    g.set_tuning(kv.split("=")[0], int(kv.split("=")[1]))
def occ(k): scan(k); g.combine_maps_occupancy()
def scan(k):
    d, n, dt, ego, tf = dev[k % len(dev)]; g.process_pointcloud_device(d.value, n, dt, ego, tf)
def timed(f, n):
    for k in range(200): f(k)
    t0 = time.perf_counter()
    for k in range(n): f(k)
    return (time.perf_counter() - t0) / n * 1e6
def sync(k): scan(k); g.combine_maps()
def at_once(k): scan(k); g.combine_maps_async().result()
pend = [None]
def piped(k):
    scan(k)
    if pend[0] is not None: pend[0].result()
    pend[0] = g.combine_maps_async()
print("%s: synchronous %.1f us | async + result at once %.1f us |" % (name, timed(sync, steps), timed(at_once, steps)), end=" ")
print("pipelined %.1f us |" % timed(piped, steps), end=" "); pend[0].result()
print("synchronous occupancy %.1f us %s" % (timed(occ, steps), " ".join(knobs)))

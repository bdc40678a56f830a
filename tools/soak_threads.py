#!/usr/bin/env python3
"""Three threads on ONE mapper for a while, as a ROS node would (gvom_ros.py:44-51): one feeds scans, one
combines (synchronous and asynchronous calls alternating), one reads debug outputs.  Looks for deadlocks, faults
and errors; results depend on the interleaving and are not compared.  Usage: tools/soak_threads.py [seconds]"""
import os, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "g-vom_amd")]
import numpy as np
import gvom, synth
secs = float(sys.argv[1]) if len(sys.argv) > 1 else 10.0
params, scans = synth.config_inputs("c2", n_scans=8)
g = gvom.Gvom(*params)
stop = time.time() + secs
counts = {"scan": 0, "combine": 0, "async": 0, "debug": 0}
errors = []

def scanner():
    k = 0
    try:
        while time.time() < stop:
            pc, ego, tf = scans[k % 8]; g.process_pointcloud(pc, ego, tf); counts["scan"] += 1; k += 1
    except Exception as e:
        errors.append(("scan", repr(e)))

def combiner():
    k = 0
    try:
        while time.time() < stop:
            if k % 3 == 2:
                p = g.combine_maps_async(); time.sleep(0.0002); p.result(); counts["async"] += 1
            elif k % 3 == 1:
                g.combine_maps_occupancy(); counts["combine"] += 1
            else:
                g.combine_maps(); counts["combine"] += 1
            k += 1
    except Exception as e:
        errors.append(("combine", repr(e)))

def reader():
    try:
        while time.time() < stop:
            g.make_debug_height_map(); g.get_map_as_occupancy_grid(); g.read_dense(0); counts["debug"] += 1
            time.sleep(0.01)
    except Exception as e:
        errors.append(("debug", repr(e)))

ts = [threading.Thread(target=f) for f in (scanner, combiner, reader)]
for t in ts: t.start()
for t in ts: t.join(secs + 60)
alive = [t.is_alive() for t in ts]
print("threads alive:", alive, "| calls:", counts, "| errors:", errors[:3])
sys.exit(1 if (any(alive) or errors) else 0)

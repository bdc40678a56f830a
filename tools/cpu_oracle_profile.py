#!/usr/bin/env python3
"""Where a step of the CPU oracle goes, per C function and for the numpy/Python rest, at a given thread count.
usage: tools/cpu_oracle_profile.py [threads ...]   (0 = the one-thread build)"""
import collections, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "g-vom_amd")]
import synth
from oracle import oracle


class Timed(object):
    def __init__(self, L):
        self.L, self.t, self.cache = L, collections.defaultdict(float), {}

    def __getattr__(self, name):
        f = getattr(self.L, name)
        if name in self.cache:
            return self.cache[name]

        def call(*a):
            t0 = time.perf_counter()
            r = f(*a)
            self.t[name] += time.perf_counter() - t0
            return r
        self.cache[name] = call
        return call


params, scans = synth.config_inputs("m256", n_scans=4)
for thr in [int(a) for a in sys.argv[1:]] or [0, 16, 64]:
    used = oracle.use_all_cores(thr > 0, threads=thr if thr > 0 else None)
    real = oracle.lib()
    timed = Timed(real)
    oracle._lib = timed                                    # every oracle call of this process goes through the timers
    g = oracle.OracleGvom(*params)
    g.reuse_buffers = True
    for k in range(2):
        pc, ego, tf = scans[k % 4]; g.process_pointcloud(pc, ego, tf); g.combine_maps()
    timed.t.clear()
    n, t0 = 0, time.perf_counter()
    while n < 3 or time.perf_counter() - t0 < 4.0:
        pc, ego, tf = scans[n % 4]; g.process_pointcloud(pc, ego, tf); g.combine_maps(); n += 1
    el = time.perf_counter() - t0
    inside = sum(timed.t.values())
    print("== %d thread(s): %.1f ms/step, %.1f ms of it inside the C oracle" % (used, el / n * 1e3, inside / n * 1e3))
    for name, v in sorted(timed.t.items(), key=lambda kv: -kv[1])[:12]:
        print("   %-36s %7.2f ms/step" % (name, v / n * 1e3))
    oracle._lib = None
oracle.use_all_cores(False)

#!/usr/bin/env python3
"""Where the cycles of a k_trace STEP go, measured inside the kernel under full load (diagnostic library, GVOM_TRACE_STEPPROF):
every 64th walking wave stamps s_memtime at the top of each of its first 32 steps, in front of the head region (LDS look-up /
add / tag store), behind it, and at the loop's end.  Prints the distribution of the three phases in shader-clock cycles.
(s_memtime is a scalar memory read: every stamp costs the wave ~100 cycles itself -- compare proportions, not absolutes.)
usage: tools/step_profile.py [config=m256] [key=value tuning ...]"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["GVOM_HIP_LIBRARY"] = os.path.join(ROOT, "g-vom_amd", "lib", "libgvom_hip_diag.so")
os.environ["GVOM_TRACE_TIMELINE"] = "1"
os.environ["GVOM_TRACE_STEPPROF"] = "1"
sys.path[:0] = [ROOT, os.path.join(ROOT, "g-vom_amd")]
import numpy as np
import bench, gvom, synth
name = sys.argv[1] if len(sys.argv) > 1 and "=" not in sys.argv[1] else "m256"
hip = bench.Hip(); hip.set_device(0)
params, scans = synth.config_inputs(name, n_scans=4)
dev = [(hip.to_device(pc), pc.shape[0], pc.dtype, ego, tf) for (pc, ego, tf) in scans]
g = gvom.Gvom(*params)
for kv in sys.argv[1:]:
    if "=" in kv:
        k, v = kv.split("="); g.set_tuning(k, int(v))
for k in range(9):
    d, n, dt, ego, tf = dev[k % 4]; g.process_pointcloud_device(d.value, n, dt, ego, tf); g.combine_maps()
lib = g._lib
lib.gvom_diag_timeline.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.POINTER(ctypes.c_int * 2)]
grid = (ctypes.c_int * 2)()
buf = np.zeros(1 << 23, np.uint64)
assert lib.gvom_diag_timeline(g._h, buf.ctypes.data_as(ctypes.c_void_p), buf.size, ctypes.byref(grid)) == 0
waves = grid[0] * grid[1] * 8
prof = buf[waves * 4 + 8: waves * 4 + 8 + (waves // 64 + 1) * 128].reshape(-1, 32, 4).astype(np.int64)
ok = (prof[:, :, 0] != 0) & (prof[:, :, 3] != 0)
arith = (prof[:, :, 1] - prof[:, :, 0])[ok]
head = (prof[:, :, 2] - prof[:, :, 1])[ok]
tail = (prof[:, :, 3] - prof[:, :, 2])[ok]
nxt = (prof[:, 1:, 0] - prof[:, :-1, 3])[ok[:, 1:] & ok[:, :-1]]
step = (prof[:, 1:, 0] - prof[:, :-1, 0])[ok[:, 1:] & ok[:, :-1]]
print("%s: %d sampled waves, %d steps" % (name, int(ok.any(1).sum()), int(ok.sum())))
for label, a in (("position .. run heads (arithmetic, lane masks, DPP)", arith), ("head region (look-up, add, tag store)", head),
                 ("length test", tail), ("loop edge to the next step's top", nxt), ("whole step (top to top)", step)):
    a = a[(a > 0) & (a < 1 << 20)]
    print("  %-52s median %6d   p10 %6d   p90 %6d   mean %8.0f cycles" % (label, np.median(a), np.percentile(a, 10), np.percentile(a, 90), a.mean()))

#!/usr/bin/env python3
"""Timeline of ONE k_trace launch from the diagnostic library (make -C g-vom_amd diag; GVOM_TRACE_TIMELINE=1):
every wave records {start, set-up done, end} (100 MHz ticks) and where it ran.  Prints, per dispatch row
(endpoint blocks, step segments), when its waves start and end and how many of them walk; the number of resident
and of walking waves over time; and the spread of busy time over the SIMDs -- i.e. where the kernel's time goes
that the instruction count does not explain (launch ramp, tail, imbalance).
usage: tools/trace_timeline.py [config=m256] [key=value tuning ...]"""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["GVOM_HIP_LIBRARY"] = os.path.join(ROOT, "g-vom_amd", "lib", "libgvom_hip_diag.so")
os.environ["GVOM_TRACE_TIMELINE"] = "1"
sys.path[:0] = [ROOT, os.path.join(ROOT, "g-vom_amd")]
import numpy as np           # noqa: E402
import bench                 # noqa: E402
import gvom                  # noqa: E402
import synth                 # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 and "=" not in sys.argv[1] else "m256"
save = [kv.split("=", 1)[1] for kv in sys.argv[1:] if kv.startswith("save=")]       # save=<file.npy>: the raw per-wave records
sys.argv = [a for a in sys.argv if not a.startswith("save=")]
hip = bench.Hip(); hip.set_device(0)
params, scans = synth.config_inputs(name, n_scans=4)
scans = (scans * 4)[:4]
dev = [(hip.to_device(pc), pc.shape[0], pc.dtype, ego, tf) for (pc, ego, tf) in scans]
g = gvom.Gvom(*params)
for kv in sys.argv[1:]:
    if "=" in kv:
        k, v = kv.split("="); g.set_tuning(k, int(v))
for k in range(41):
    d, n, dt, ego, tf = dev[k % 4]; g.process_pointcloud_device(d.value, n, dt, ego, tf); g.combine_maps()
lib = g._lib
lib.gvom_diag_timeline.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.POINTER(ctypes.c_int * 2)]
grid = (ctypes.c_int * 2)()
buf = np.zeros(1 << 22, np.uint64)
rc = lib.gvom_diag_timeline(g._h, buf.ctypes.data_as(ctypes.c_void_p), buf.size, ctypes.byref(grid))
assert rc == 0, rc
gx, gy = grid[0], grid[1]
w = buf[:gx * gy * 8 * 4].reshape(gy, gx * 8, 4).astype(np.int64)
modes = buf[gx * gy * 8 * 4:gx * gy * 8 * 4 + 3]
if save:
    np.save(save[0], w)              # [row][workgroup * 8 + wave][start, set-up done, end, where]; the scan is scans[0] of synth.config_inputs(name, 4)
print("run steps by lookup mode (upper bounds: a run may end early): integer + window test %d, literal f64 %d, integer without window test %d"
      % (modes[0], modes[1], modes[2]))
started = w[:, :, 0] > 0
t0 = w[:, :, 0][started].min()
us = lambda t: (t - t0) / 100.0
print("%s: grid %d x %d workgroups of 8 waves; kernel span (first wave start -> last wave end) %.1f us"
      % (name, gx, gy, us(w[:, :, 2][started].max())))
print("row  waves  walk   start us (min / median / max)   end us (median / max)   walking waves: duration median / max")
for r in range(gy):
    m = started[r]
    st, en, su = us(w[r, m, 0]), us(w[r, m, 2]), w[r, m, 1]
    walk = su > 0
    dur = (w[r, m, 2] - w[r, m, 0])[walk] / 100.0
    print("%3d  %5d  %5d   %6.1f / %6.1f / %6.1f           %6.1f / %6.1f         %s"
          % (r, m.sum(), walk.sum(), st.min(), np.median(st), st.max(), np.median(en), en.max(),
             "%.1f / %.1f (set-up %.1f)" % (np.median(dur), dur.max(), np.median((w[r, m, 1] - w[r, m, 0])[walk]) / 100.0) if walk.any() else "-"))
# occupancy over time
S, E, U = us(w[:, :, 0][started]), us(w[:, :, 2][started]), w[:, :, 1][started] > 0
end = E.max()
print("time us : resident waves / walking waves (of 8192 wave slots)")
for t in np.arange(0.0, end + 1.0, 2.0):
    res = ((S <= t) & (E > t)).sum()
    wk = ((S <= t) & (E > t) & U).sum()
    print("  %5.1f : %5d / %5d  %s" % (t, res, wk, "#" * int(wk / 128)))
# per-SIMD busy time (walking waves only): HW_ID bits: wave 3:0, simd 5:4, pipe 7:6, cu 11:8, sh 12, se 15:13 (gfx9); XCC_ID 3:0
hw = w[:, :, 3][started]
simd = (hw >> 4) & 3; cu = (hw >> 8) & 15; se = (hw >> 13) & 7; xcc = (hw >> 32) & 15
key = ((xcc * 8 + se) * 16 + cu) * 4 + simd
busy = np.zeros(int(key.max()) + 1)
np.add.at(busy, key[U], (E - S)[U])
nz = busy[busy > 0]
print("SIMDs with walking waves: %d; wave-microseconds per SIMD: min %.1f / median %.1f / max %.1f (sum %.0f)"
      % (nz.size, nz.min(), np.median(nz), nz.max(), nz.sum()))
lastend = np.zeros(int(key.max()) + 1)
np.maximum.at(lastend, key, E)
le = lastend[lastend > 0]
print("last wave end per SIMD: 10%% %.1f / median %.1f / 90%% %.1f / max %.1f us" % tuple(np.percentile(le, [10, 50, 90, 100])))

# where the dispatcher puts the workgroups: linear workgroup id -> (XCC, SE, CU); do ids g and g + 256 share a CU?
wg_hw = w[:, :, 3].reshape(gy * gx, 8)[:, 0]
nwg = gy * gx
order = np.argsort((wg_hw >> 40) & 0xffffff, kind="stable")      # logical workgroup ids in DISPATCH order
wg_hw = wg_hw[order]
wg_cu = (((wg_hw >> 32) & 15) * 8 + ((wg_hw >> 13) & 7)) * 16 + ((wg_hw >> 8) & 15)
wg_xcc = (wg_hw >> 32) & 15
first = min(nwg, 1024)
print("first %d workgroups: XCC == id %% 8 for %d of them; distinct CUs %d; workgroups per CU min %d / max %d"
      % (first, int((wg_xcc[:first] == np.arange(first) % 8).sum()), len(set(wg_cu[:first].tolist())),
         np.bincount(np.unique(wg_cu[:first], return_inverse=True)[1]).min(), np.bincount(np.unique(wg_cu[:first], return_inverse=True)[1]).max()))
for d in (8, 64, 128, 256, 512):
    if nwg > d:
        k = min(first, nwg - d)
        print("  ids g and g + %d on the same CU: %d of %d" % (d, int((wg_cu[:k] == wg_cu[d:d + k]).sum()), k))
print("  CU of workgroups 0..31:", [int(c) for c in wg_cu[:32]])
print("  CU of workgroups 256..287:", [int(c) for c in wg_cu[256:288]])
# alive workgroups per CU (a workgroup is alive if any of its waves walks)
alive_wg = (w[:, :, 1].reshape(gy * gx, 8) > 0).any(1)[order]
print("walking workgroups among the first 1024 dispatched: %d of %d" % (int(alive_wg[:1024].sum()), int(alive_wg.sum())))
cus, inv = np.unique(wg_cu, return_inverse=True)
per_cu = np.bincount(inv, weights=alive_wg.astype(float))
print("walking workgroups per CU: min %d / median %d / max %d of %d CUs" % (per_cu.min(), np.median(per_cu), per_cu.max(), cus.size))

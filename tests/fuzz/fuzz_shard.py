#!/usr/bin/env python3
"""Slab-sharded scan + fusion kernels against the unsharded handle on ONE GPU: for every seed of
tests/test_hip_parity.py::_fuzz_case (grid size rounded to a multiple of the world size), W sharded
handles are fed the whole cloud and the rows each owns must equal the unsharded handle's, for every
ring slot after every scan and for the fused map after every combine (the 2-D stage needs the
collective and is covered by tests/test_hip_sharded.py).  Usage: tests/fuzz/fuzz_shard.py <first> <count>"""
import os, sys, io, contextlib, ctypes
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in ("tests", "tests/golden", "g-vom_amd", ""):
    sys.path.insert(0, os.path.join(ROOT, p))
import importlib
thp = importlib.import_module("test_hip_parity")
import gvom

def owned_rows_mask(origin_y, xy, r, W):
    om = int(origin_y) % xy
    sy = (np.arange(xy) + om) % xy                      # storage row of window row y
    rows = xy // W
    return (sy >= r * rows) & (sy < (r + 1) * rows)

def merged(handles, which, xy, zs, W):
    out = None
    for r, h in enumerate(handles):
        d = h.read_dense(which)
        if d is None:
            return None
        if out is None:
            out = [np.array(a, copy=True) for a in d[:4]]
        m = owned_rows_mask(d[4][1], xy, r, W)
        m3 = np.broadcast_to(m[None, :, None], (zs, xy, xy)).reshape(-1)
        for k in range(4):
            out[k][m3] = d[k][m3]
    return out

first, count = int(sys.argv[1]), int(sys.argv[2])
bad = []
for seed in range(first, first + count):
    W = (2, 4, 8)[seed % 3]
    params, steps = thp._fuzz_case(seed)
    xy = max(W, (params[2] // W) * W)
    params = params[:2] + (xy,) + params[3:]
    zs = params[3]
    try:
        with contextlib.redirect_stdout(io.StringIO()):
            g0 = gvom.Gvom(*params)
            hs = [gvom.Gvom(*params, _shard=(r, W)) for r in range(W)]
            for st in steps:
                if st[0] == "scan":
                    g0.process_pointcloud(*st[1:])
                    for h in hs:
                        h.process_pointcloud(*st[1:])
                    b = g0.last_buffer_index
                    want = g0.read_dense(b)
                    got = merged(hs, b, xy, zs, W)
                    assert (want is None) == (got is None), "slot presence"
                    if want is not None:
                        for k, nm in enumerate(("state", "hit", "total", "minh")):
                            wk, gk = want[k], got[k]
                            if nm == "state":            # rows are numbered per handle: compare the classes
                                assert np.array_equal(np.where(wk >= 0, 0, wk), np.where(gk >= 0, 0, gk)), "slot state"
                            else:
                                assert np.array_equal(wk, gk), "slot " + nm
                else:
                    r0 = g0.combine_maps()
                    rcs = [h._lib.gvom_combine_fuse(h._h, None) for h in hs]
                    for h in hs:
                        h._lib.gvom_sync(h._h)
                    if r0 is None:
                        assert all(rc == gvom.GVOM_EMPTY_BUFFER for rc in rcs), "empty ring"
                        continue
                    want = g0.read_dense(gvom.GVOM_WHICH_FUSED)
                    got = merged(hs, gvom.GVOM_WHICH_FUSED, xy, zs, W)
                    for k, nm in enumerate(("state", "hit", "total", "minh")):
                        wk, gk = want[k], got[k]
                        if nm == "state":
                            assert np.array_equal(np.where(wk >= 0, 0, wk), np.where(gk >= 0, 0, gk)), "fused state"
                        else:
                            assert np.array_equal(wk, gk), "fused " + nm
    except AssertionError as e:
        bad.append((seed, W, str(e)[:80]))
print("checked %d seeds, %d failures" % (count, len(bad)))
for b in bad[:20]:
    print("  seed %d world %d: %s" % b)

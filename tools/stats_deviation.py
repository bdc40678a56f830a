#!/usr/bin/env python3
"""How far the per-voxel statistics (eigenvalue columns of make_debug_voxel_map) are from the rows the REFERENCE produced, on the
golden fixtures F1-F6: max absolute and relative deviation.  (The reference accumulates two centred passes with f64 atomics, this
library own-voxel raw moments + shift algebra: different algorithms, compared by tolerance.)  Usage: tools/stats_deviation.py"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ("tests", "tests/golden", "g-vom_amd", ""):
    sys.path.insert(0, os.path.join(ROOT, p))
import scenarios, gvom
G = os.path.join(ROOT, "tests", "golden")
for name in ("f1", "f2", "f3", "f4", "f5", "f6"):
    want = np.load(os.path.join(G, name + ".npz"))
    sc = scenarios.scenario_from_record(want)
    got = scenarios.run_and_record(lambda *p: gvom.Gvom(*p, voxel_statistics=True), sc, record_debug=True)
    worst_abs = worst_rel = 0.0; n = 0
    for k in want.files:
        if k.endswith("debug_voxel_map") and k in got:
            a, b = np.asarray(got[k])[:, 5:].astype(np.float64), np.asarray(want[k])[:, 5:].astype(np.float64)
            ok = np.isfinite(a) & np.isfinite(b)
            d = np.abs(a - b)[ok]
            worst_abs = max(worst_abs, float(d.max()) if d.size else 0.0)
            rel = (d / np.maximum(np.abs(b[ok]), 1e-3))
            worst_rel = max(worst_rel, float(rel.max()) if rel.size else 0.0)
            n += int(ok.sum())
            assert np.array_equal(np.isnan(a), np.isnan(b))
    print(name, "eigen columns: %d values, max |d| %.3g, max |d| / max(|ref|, 1e-3) %.3g" % (n, worst_abs, worst_rel))

# the same at FULL size against the CPU oracle (c2 grid, three 131,072-point scans, buffer 2): float32 merges over the ring,
# hundreds of returns per voxel
if "c2" in sys.argv[1:]:
    from oracle import oracle
    import synth
    params, scans = synth.config_inputs("c2", n_scans=3)
    params = params[:4] + (2,) + params[5:]
    steps = []
    for sc_ in scans:
        steps += [("scan",) + sc_, ("combine",)]
    sc = {"params": params, "steps": steps}
    want = scenarios.run_and_record(lambda *p: oracle.OracleGvom(*p, voxel_statistics=True), sc, record_debug=True)
    got = scenarios.run_and_record(lambda *p: gvom.Gvom(*p, voxel_statistics=True), sc, record_debug=True)
    for k in sorted(want):
        if k.endswith("debug_voxel_map") and k in got:
            a, b = np.asarray(got[k])[:, 5:].astype(np.float64), np.asarray(want[k])[:, 5:].astype(np.float64)
            ok = np.isfinite(a) & np.isfinite(b)
            d = np.abs(a - b)[ok]
            rel = d / np.maximum(np.abs(b[ok]), 1e-3)
            print("c2 vs oracle", k, "%d values, max |d| %.3g, max |d| / max(|ref|, 1e-3) %.3g, 99.9th percentile |d| %.3g" % (int(ok.sum()), float(d.max()), float(rel.max()), float(np.percentile(d, 99.9))))

#!/usr/bin/env python3
"""Runs tests/test_hip_parity.py::_fuzz_case for a range of seeds on the GPU box and reports the
seeds whose HIP records differ from the oracle.  Usage: tests/fuzz/fuzz_many.py <first> <count> [stats|ondemand] [ilv=K] [p2] [eager] [dirsort]
(ilv=K: the HIP mapper traces with the sub-cloud interleave forced to K -- a permutation of who traces which return; it applies
to the scans whose length K divides; p2: grid sizes snapped to powers of two, z_size <= xy_size -- the grids on which k_trace takes
its mask-wrap and no-window-test step bodies)"""
import os, sys, io, contextlib
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in ("tests", "tests/golden", "g-vom_amd", ""):
    sys.path.insert(0, os.path.join(ROOT, p))
import importlib
thp = importlib.import_module("test_hip_parity")
import scenarios, gvom
from parity import compare_records
from oracle import oracle
first, count = int(sys.argv[1]), int(sys.argv[2])
stats = "stats" in sys.argv[3:]
ilv = [int(a[4:]) for a in sys.argv[3:] if a.startswith("ilv=")]


def hip_mapper(*p):
    # (default: no statistics at any step -- the north-star path, eager fusion included; "ondemand": the class default, statistics for as
    # long as the recorded debug reads ask for them)
    g = gvom.Gvom(*p, voxel_statistics=True) if stats else (gvom.Gvom(*p) if "ondemand" in sys.argv[3:] else gvom.Gvom(*p, voxel_statistics=False))
    if ilv:
        g.set_tuning("interleave", ilv[0])
    for a_ in sys.argv[3:]:
        if a_.startswith("dirsort"):         # every scan of >= 256 returns traced in directional order (k_dirbin_*); dirsort=2: elevation rows
            g.set_tuning("dirsort", int(a_[8:]) if a_.startswith("dirsort=") else 1)
    return g
bad = []
for seed in range(first, first + count):
    params, steps = thp._fuzz_case(seed)
    if "p2" in sys.argv[3:]:
        xy2 = 1 << max(2, int(round(np.log2(max(params[2], 4)))))
        zs2 = min(xy2, 1 << max(0, int(round(np.log2(max(params[3], 1))))))
        params = params[:2] + (xy2, zs2) + params[4:]
    if "eager" in sys.argv[3:]:
        # one-slot rings on grids the eager fusion takes (buffer_size 1, xy_size % 16 == 0, z_size >= 4): scans with and without
        # a combine behind them, rejected and empty scans, moving windows -- k_encfuse, its adoption and its fall-backs
        params = params[:2] + (16 * max(1, min(4, round(params[2] / 16))), max(4, params[3]), 1) + params[5:]
    sc = {"params": params, "steps": steps}
    try:
        with contextlib.redirect_stdout(io.StringIO()):
            want = scenarios.run_and_record((lambda *p: oracle.OracleGvom(*p, voxel_statistics=True)) if stats else oracle.OracleGvom, sc)
            got = scenarios.run_and_record(hip_mapper, sc)
        compare_records(got, want, float_tol=1e-5, **({"stats_rtol": 1e-4, "stats_atol": 2e-5} if stats else {}))
    except AssertionError as e:
        bad.append((seed, str(e)[:100]))
print("checked %d seeds, %d failures" % (count, len(bad)))
for b in bad[:20]:
    print("  seed", b[0], b[1])

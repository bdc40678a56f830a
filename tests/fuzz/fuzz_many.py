#!/usr/bin/env python3
"""Runs tests/test_hip_parity.py::_fuzz_case for a range of seeds on the GPU box and reports the
seeds whose HIP records differ from the oracle.  Usage: tests/fuzz/fuzz_many.py <first> <count> [stats]"""
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
stats = len(sys.argv) > 3
bad = []
for seed in range(first, first + count):
    params, steps = thp._fuzz_case(seed)
    sc = {"params": params, "steps": steps}
    try:
        with contextlib.redirect_stdout(io.StringIO()):
            want = scenarios.run_and_record((lambda *p: oracle.OracleGvom(*p, voxel_statistics=True)) if stats else oracle.OracleGvom, sc)
            got = scenarios.run_and_record((lambda *p: gvom.Gvom(*p, voxel_statistics=True)) if stats else gvom.Gvom, sc)
        compare_records(got, want, float_tol=1e-5, **({"stats_rtol": 1e-4, "stats_atol": 2e-5} if stats else {}))
    except AssertionError as e:
        bad.append((seed, str(e)[:100]))
print("checked %d seeds, %d failures" % (count, len(bad)))
for b in bad[:20]:
    print("  seed", b[0], b[1])

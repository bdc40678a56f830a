#!/usr/bin/env python3
"""Debug helper: prints / dumps one case of tests/test_hip_parity.py::_fuzz_case and (on a GPU box)
the voxels where the HIP path and the oracle disagree.  Usage: tests/fuzz/fuzz_case.py <seed> [hip]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in ("tests", "tests/golden", "g-vom_amd", ""):
    sys.path.insert(0, os.path.join(ROOT, p))
import importlib
thp = importlib.import_module("test_hip_parity")
import scenarios
from oracle import oracle

seed = int(sys.argv[1])
params, steps = thp._fuzz_case(seed)
print("params", params)
for s in steps:
    print(" ", s[0], (s[1].shape, s[1].dtype.name, s[2], "tf" if s[3] is not None else None) if s[0] == "scan" else "")
sc = {"params": params, "steps": steps}
want = scenarios.run_and_record(oracle.OracleGvom, sc)
if len(sys.argv) > 2:
    import gvom
    got = scenarios.run_and_record(gvom.Gvom, sc)
    xy, zs = params[2], params[3]
    for k in sorted(want):
        if k in got and isinstance(want[k], np.ndarray) and want[k].shape == np.asarray(got[k]).shape and want[k].dtype.kind in "iu":
            a, b = np.asarray(got[k]), want[k]
            if not np.array_equal(a, b):
                idx = np.nonzero(a.ravel() != b.ravel())[0]
                print(k, "differs in", idx.size, "places")
                for i in idx[:12]:
                    z, r = divmod(int(i), xy * xy); y, x = divmod(r, xy)
                    print("   idx %d (x %d y %d z %d): hip %d oracle %d" % (i, x, y, z, a.ravel()[i], b.ravel()[i]))
np.savez("/tmp/fuzz_%d.npz" % seed, **{k: v for k, v in want.items() if isinstance(v, np.ndarray)})

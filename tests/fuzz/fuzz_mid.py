#!/usr/bin/env python3
"""Mid-size randomized parity run (not in the test-suite: ~0.5 s of CPU oracle per case): grids of
64..160 x 8..64 voxels, lidar-shaped scans of a random box scene with random sensor poses, random
thresholds, buffer 1..8, 2..8 scans with interleaved combines.  Usage: tests/fuzz/fuzz_mid.py <first> <count>"""
import os, sys, io, contextlib
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in ("tests", "tests/golden", "g-vom_amd", ""):
    sys.path.insert(0, os.path.join(ROOT, p))
import scenarios, gvom, synth
from parity import compare_records
from oracle import oracle

def case(seed):
    rng = np.random.default_rng(seed)
    xy_res = float(rng.choice([0.15, 0.2, 0.4])); z_res = float(rng.choice([0.1, 0.2, 0.3]))
    xy = int(rng.integers(64, 161)); zs = int(rng.integers(8, 65)); buf = int(rng.integers(1, 9))
    params = (xy_res, z_res, xy, zs, buf, float(rng.choice([0.0, 1.0, 2.5])),
              float(rng.uniform(0.2, 0.8)), float(rng.uniform(0.2, 0.8)), float(rng.uniform(0.1, 0.5)),
              float(rng.uniform(1.0, 3.0)), float(rng.uniform(0.5, 5.0)), float(rng.uniform(0.3, 1.5)), 1, 1)
    scene = synth.make_scene(int(rng.integers(0, 1000)), n_boxes=int(rng.integers(3, 25)), extent=xy * xy_res * 0.6)
    steps = []
    pos = rng.uniform(-2, 2, 3) * np.array([1, 1, 0.2])
    for k in range(int(rng.integers(2, 9))):
        pos = pos + rng.uniform(-1.0, 1.0, 3) * np.array([1, 1, 0.05])
        beams = int(rng.choice([8, 16, 32])); az = int(rng.choice([256, 512, 1024]))
        pc = synth.lidar_scan(scene, beams=beams, azimuths=az, sensor=tuple(pos), yaw=float(rng.uniform(0, 6.28)),
                              noise_seed=int(rng.integers(0, 1 << 30)), dtype=rng.choice([np.float32, np.float64]))
        tf = scenarios.rot_z(float(rng.uniform(-0.05, 0.05)), tuple(rng.uniform(-0.05, 0.05, 3))) if rng.random() < 0.3 else None
        steps.append(("scan", pc, tuple(float(v) for v in pos), tf))
        if rng.random() < 0.6:
            steps.append(("combine",))
    steps.append(("combine",))
    return params, steps

first, count = int(sys.argv[1]), int(sys.argv[2])
bad = []
for seed in range(first, first + count):
    params, steps = case(seed)
    sc = {"params": params, "steps": steps}
    try:
        with contextlib.redirect_stdout(io.StringIO()):
            want = scenarios.run_and_record(oracle.OracleGvom, sc, record_debug=(seed % 4 == 0))
            # (every third case through the class default -- statistics on demand --, the others without statistics at any step)
            got = scenarios.run_and_record(gvom.Gvom if seed % 3 == 0 else (lambda *p: gvom.Gvom(*p, voxel_statistics=False)), sc, record_debug=(seed % 4 == 0))
        compare_records(got, want, float_tol=1e-5)
    except AssertionError as e:
        bad.append((seed, str(e)[:100]))
print("checked %d seeds, %d failures" % (count, len(bad)))
for b in bad[:20]:
    print("  seed", b[0], b[1])

#!/usr/bin/env python3
"""Localises sharded-path failures: (A) ranks as threads on a mid grid, (B) a fuzz seed. Progress after every step."""
import os, sys, io, contextlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ("tests", "tests/golden", "tests/fuzz", "g-vom_amd", ""):
    sys.path.insert(0, os.path.join(ROOT, p))
import numpy as np
import gvom, gvom_sharded, synth
from shard_threads import run_ranks

def say(*a):
    print(*a, flush=True)

mode = sys.argv[1]
if mode == "A":
    W = 2
    params = (0.2, 0.2, 64, 16, 2) + synth.REF_TAIL
    rng = np.random.default_rng(1)
    pc = (rng.uniform(-5, 5, (4000, 3)) * np.array([1, 1, 0.2])).astype(np.float32)
    shares = [pc[:1500], pc[1500:]]
    def body(r, sh):
        for k in range(3):
            say("rank", r, "scan", k)
            sh.process_pointcloud(shares[r], (0.1 * k, -0.2, 0.05))
            sh.b.sync()
            say("rank", r, "scan done", k)
            out = sh.combine_maps()
            say("rank", r, "combine done", k, out is not None)
        return 1
    say(run_ranks(W, params, body))
else:
    import fuzz_shard
    seed = int(sys.argv[2])
    params, steps = fuzz_shard.thp._fuzz_case(seed)
    say("seed", seed, "params", params, "steps", [(s[0], np.asarray(s[1]).shape if s[0] == "scan" else None) for s in steps])
    say(fuzz_shard.check_case(seed))
say("done", mode)

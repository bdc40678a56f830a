#!/usr/bin/env python3
"""Sharded map against the unsharded handle on ONE GPU: for every seed of
tests/test_hip_parity.py::_fuzz_case (grid size rounded to a multiple of 4 x the world size), W
ranks run as threads (tests/shard_threads.py), each fed a RAGGED share of every cloud (shares differ
in length, some are empty); after every scan the rows each rank owns of the ring slot, and after every
combine those of the fused map, must equal the unsharded handle's, and every rank's returned 2-D maps
must equal the unsharded ones.  Usage: tests/fuzz/fuzz_shard.py <first> <count>"""
import os, sys, io, contextlib
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in ("tests", "tests/golden", "g-vom_amd", ""):
    sys.path.insert(0, os.path.join(ROOT, p))
import importlib
thp = importlib.import_module("test_hip_parity")
import gvom
from shard_threads import run_ranks


def owned_rows_mask(origin_y, xy, r, W):
    om = int(origin_y) % xy
    sy = (np.arange(xy) + om) % xy                      # storage row of window row y
    rows = xy // W
    return (sy >= r * rows) & (sy < (r + 1) * rows)


def _where(k, a, b, origin_world, params, W, rank):
    """which returned array differs, in how many cells, and which ranks OWN the storage rows of those cells
    (an all-gather that arrived late shows up as whole foreign slabs; a kernel race as scattered cells)"""
    name = ("origin", "positive", "negative", "roughness", "visibility")[k]
    if a.shape != b.shape or a.ndim != 2:
        return "%s on rank %d" % (name, rank)
    xy = params[2]
    bad = np.argwhere(~((a == b) | ((a != a) & (b != b))))
    oy = int(round(origin_world[1] / params[0]))
    owners = sorted(set((((bad[:, 1] + oy) % xy) // (xy // W)).tolist()))
    return "%s on rank %d: %d cells, window rows y %d..%d, owner ranks %s" % (
        name, rank, bad.shape[0], bad[:, 1].min(), bad[:, 1].max(), owners)


def shares_of(pc, W, rng):
    """ragged split of a cloud: random cut points, so shares differ in length and may be empty"""
    n = pc.shape[0]
    cuts = np.sort(rng.integers(0, n + 1, W - 1)) if n else np.zeros(W - 1, np.int64)
    idx = np.concatenate([[0], cuts, [n]])
    return [pc[idx[r]:idx[r + 1]] for r in range(W)]


def check_case(seed):
    W = (2, 4, 8)[seed % 3]
    params, steps = thp._fuzz_case(seed)
    xy = max(4 * W, (params[2] // (4 * W)) * 4 * W)
    params = params[:2] + (xy,) + params[3:]
    zs = params[3]
    rng = np.random.default_rng(seed)
    plan = []
    for st in steps:
        if st[0] == "scan":
            plan.append(("scan", shares_of(np.asarray(st[1]), W, rng), st[2], st[3]))
        else:
            plan.append(("combine",))
    verbose = bool(os.environ.get("GVOM_FUZZ_VERBOSE"))
    say = (lambda *a: print(*a, file=sys.stderr, flush=True)) if verbose else (lambda *a: None)
    say("world", W, "params", params)
    g0 = gvom.Gvom(*params)
    want = []
    for st, pl in zip(steps, plan):
        if st[0] == "scan":
            say("unsharded scan", np.asarray(st[1]).shape)
            g0.process_pointcloud(*st[1:])
            b = g0.last_buffer_index
            want.append(("scan", b, g0.read_dense(b), g0.buffer_index))
        else:
            say("unsharded combine")
            out = g0.combine_maps()
            want.append(("combine", out, g0.read_dense(gvom.GVOM_WHICH_FUSED) if out is not None else None,
                         g0.combined_cell_count_cpu))

    def body(r, sh):
        for pl, wt in zip(plan, want):
            if pl[0] == "scan":
                say("rank", r, "scan", pl[1][r].shape)
                sh.process_pointcloud(pl[1][r], pl[2], pl[3])
                sh.b.sync()
                say("rank", r, "scan done")
                assert sh.b.g.buffer_index == wt[3], "ring index"
                if wt[2] is None:
                    continue
                got = sh.b.g.read_dense(wt[1])
                assert got is not None, "slot presence"
                m = np.broadcast_to(owned_rows_mask(got[4][1], xy, r, W)[None, :, None], (zs, xy, xy)).reshape(-1)
                for k, nm in enumerate(("state", "hit", "total", "minh")):
                    a, b = wt[2][k][m], got[k][m]
                    if nm == "state":                    # rows are numbered per handle: compare the classes
                        a, b = np.where(a >= 0, 0, a), np.where(b >= 0, 0, b)
                    assert np.array_equal(a, b), "slot " + nm
            else:
                say("rank", r, "combine")
                out = sh.combine_maps()
                say("rank", r, "combine done")
                assert (out is None) == (wt[1] is None), "combine presence"
                if out is None:
                    continue
                for k, (a, b) in enumerate(zip(out, wt[1])):
                    assert a.dtype == b.dtype and np.array_equal(a, b), "returned maps: " + _where(k, a, b, wt[1][0], params, W, r)
                assert sh.combined_cell_count_cpu == wt[3], "cell count"
                got = sh.b.g.read_dense(gvom.GVOM_WHICH_FUSED)
                m = np.broadcast_to(owned_rows_mask(got[4][1], xy, r, W)[None, :, None], (zs, xy, xy)).reshape(-1)
                for k, nm in enumerate(("state", "hit", "total", "minh")):
                    a, b = wt[2][k][m], got[k][m]
                    if nm == "state":
                        a, b = np.where(a >= 0, 0, a), np.where(b >= 0, 0, b)
                    assert np.array_equal(a, b), "fused " + nm
        return True

    run_ranks(W, params, body)
    return W


if __name__ == "__main__":
    first, count = int(sys.argv[1]), int(sys.argv[2])
    bad = []
    for seed in range(first, first + count):
        try:
            with contextlib.redirect_stdout(io.StringIO()):
                check_case(seed)
        except AssertionError as e:
            bad.append((seed, (2, 4, 8)[seed % 3], str(e)[:200]))
        if (seed - first + 1) % 200 == 0:                   # (a long campaign shows that it is alive)
            print("... %d seeds, %d failures so far" % (seed - first + 1, len(bad)), file=sys.stderr, flush=True)
    print("checked %d seeds, %d failures" % (count, len(bad)))
    for b in bad[:20]:
        print("  seed %d world %d: %s" % b)

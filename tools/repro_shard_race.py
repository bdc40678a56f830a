#!/usr/bin/env python3
"""the sharded-vs-unsharded difference the driver saw once (fuzz_shard seed 70070,
8 ranks, "returned maps") and the builder's leases never did.  Runs one fuzz_shard seed over and over in
ONE process with the ranks' device data moved either way:

  legacy   round 2's ThreadComm for the combine's all-gather: hipMemcpy(device -> device) on the NULL stream,
           no wait.  The handles' streams are hipStreamNonBlocking (not ordered against the null stream) and a
           device-to-device hipMemcpy returns before it has run (tools/d2d_probe.hip: 3 us for a 0.9 ms copy),
           so k_map2d can overtake its input.
  stream   this round's ThreadComm: hipMemcpyAsync on the receiving handle's stream (the ordering RCCL
           gives), drained before the senders may overwrite their regions.

  +backlog (legacy+backlog / stream+backlog): every transfer is preceded by a 256 MiB device-to-device hipMemcpy
           on the null stream (~0.2 ms of copy-engine backlog in front of whatever else uses that stream).  The
           tiny copies of the fuzz cases normally win the race against the consumer's launch; behind a backlog
           they lose it every time -- which is what a loaded box does to them now and then.

usage: tools/repro_shard_race.py <legacy|stream>[+backlog] <first seed> <repeats> [seeds per repeat = 1]
(the driver's failure came in the 71st case of the campaign 70000..70089: `legacy 70000 5 90` re-runs that
sequence).  Prints every mismatch with the rank, the array and the owner ranks of the differing rows."""
import contextlib
import ctypes
import io
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ("tests", "tests/golden", "tests/fuzz", "g-vom_amd", ""):
    sys.path.insert(0, os.path.join(ROOT, p))
import gvom_sharded          # noqa: E402
import shard_threads         # noqa: E402
import fuzz_shard            # noqa: E402


def patch_transport(legacy, backlog):
    rt = ctypes.CDLL("libamdhip64.so")
    rt.hipMemcpy.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
    rt.hipMalloc.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_size_t]
    big = [ctypes.c_void_p(), ctypes.c_void_p()]
    nbig = 256 << 20
    if backlog:
        for b in big:
            assert rt.hipMalloc(ctypes.byref(b), nbig) == 0
    stream_copy = shard_threads.ThreadComm._copy
    stream_drain = shard_threads.ThreadComm._drain
    stream_allgather = shard_threads.ThreadComm.allgather_rows
    import threading
    tl = threading.local()

    # Only the COMBINE's all-gather is run the round-2 way: a consumer that overtakes it reads stale float64 heights
    # (wrong maps, the driver's symptom).  The scan's exchange carries voxel INDICES: overtaken, k_unpack_eps would
    # index with stale or uninitialised words and could fault the GPU -- it stays stream-ordered in every mode.
    def copy(self, backend, dst, src, nbytes):
        if not nbytes:
            return
        gather = getattr(tl, "gather", False)
        if backlog and gather:
            rt.hipMemcpy(big[0], big[1], nbig, 3)          # returns at once (tools/d2d_probe.hip); the null stream is busy for ~0.2 ms
        if legacy and gather:
            if rt.hipMemcpy(ctypes.c_void_p(dst), ctypes.c_void_p(src), nbytes, 3) != 0:
                raise RuntimeError("hipMemcpy failed")
        else:
            stream_copy(self, backend, dst, src, nbytes)

    def drain(self, backend):
        if not (legacy and getattr(tl, "gather", False)):
            stream_drain(self, backend)

    def allgather(self, backend):
        tl.gather = True
        try:
            stream_allgather(self, backend)
        finally:
            tl.gather = False

    shard_threads.ThreadComm._copy = copy
    shard_threads.ThreadComm._drain = drain
    shard_threads.ThreadComm.allgather_rows = allgather


def main():
    mode, seed0, reps = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
    span = int(sys.argv[4]) if len(sys.argv) > 4 else 1
    base, _, extra = mode.partition("+")
    if base not in ("legacy", "stream") or extra not in ("", "backlog"):
        raise SystemExit(__doc__)
    patch_transport(base == "legacy", extra == "backlog")
    bad, t0 = 0, time.time()
    n = 0
    for k in range(reps):
        for seed in range(seed0, seed0 + span):
            n += 1
            try:
                with contextlib.redirect_stdout(io.StringIO()):
                    fuzz_shard.check_case(seed)
            except AssertionError as e:
                bad += 1
                print("repeat %d seed %d: %s" % (k, seed, str(e)[:300]), flush=True)
            if n % 100 == 0:
                print("... %d cases, %d mismatches, %.0f s" % (n, bad, time.time() - t0), flush=True)
    print("%s transport, seeds %d..%d x %d: %d cases, %d mismatches" % (mode, seed0, seed0 + span - 1, reps, n, bad))
    return 0


if __name__ == "__main__":
    sys.exit(main())

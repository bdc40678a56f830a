#!/usr/bin/env python3
"""RCCL loopback on a one-GPU box: `world` thread-ranks of one process, the product's communicator
(gvom_sharded.RcclComm transport="loopback": every rank a 1-rank RCCL communicator; grouped ncclSend / ncclRecv to
itself on the receiving handle's stream; in-place ncclAllGather), the library's one-call scan and combine, checked against
the unsharded handle.  Prints the RCCL calls every rank issued.  Run it under rocprofv3 --kernel-trace --stats to see
RCCL's kernels between k_trace / k_pack and k_unpack_* (profiles/r6_rccl_loopback.txt).
Usage: tools/rccl_loopback_probe.py [worlds, e.g. 2,4] [scans]"""
import contextlib
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ("tests", "tests/golden", "g-vom_amd", ""):
    sys.path.insert(0, os.path.join(ROOT, p))
import gvom            # noqa: E402
import synth           # noqa: E402
from shard_threads import run_ranks    # noqa: E402


def main():
    worlds = [int(w) for w in (sys.argv[1] if len(sys.argv) > 1 else "2,4").split(",")]
    n_scans = int(sys.argv[2]) if len(sys.argv) > 2 else 6
    params = (0.2, 0.2, 256, 64, 2) + synth.REF_TAIL
    scene = synth.make_scene(2)
    bad = 0
    for W in worlds:
        scans = []
        for k in range(n_scans):
            ego = (0.4 * k, -0.3 * k, 0.02 * k)
            scans.append(([synth.lidar_scan(scene, beams=64, sensor=ego, yaw=2 * np.pi / 2048 * r / W, noise_seed=100 * k + r)
                           for r in range(W)], ego))
        g0 = gvom.Gvom(*params)
        want = []
        for shares, ego in scans:
            g0.process_pointcloud(np.concatenate(shares, 0), ego)
            want.append(g0.combine_maps())
        del g0
        info = [None] * W

        def body(r, sh):
            nbad, t0 = 0, time.perf_counter()
            for (shares, ego), wout in zip(scans, want):
                sh.process_pointcloud(shares[r], ego)
                out = sh.combine_maps()
                nbad += sum(0 if np.array_equal(a, c) else 1 for a, c in zip(out, wout))
            info[r] = (sh.comm.wire_stats(), sh.comm.info(), sh.last_exchange_bytes, (time.perf_counter() - t0) / len(scans) * 1e3)
            return nbad

        with contextlib.redirect_stdout(sys.stderr):
            res = run_ranks(W, params, body, transport="loopback")
        bad += sum(res)
        for r in range(W):
            w, i, last, ms = info[r]
            print("world %d rank %d: transport %s, RCCL communicator of %s rank(s) on %s; %d ncclSend/ncclRecv calls, %.2f MB, %d groups, "
                  "%d ncclAllGather; last scan sent %d B received %d B; %.2f ms per scan+combine"
                  % (W, r, i["transport"], i["rccl_comm_count"], i["pci_bus_id"], w["p2p_calls"], w["p2p_bytes"] / 1e6, w["groups"],
                     w["allgathers"], last[0], last[1], ms))
        print("world %d: %d scans + combines, maps %s the unsharded handle's" % (W, n_scans, "EQUAL" if sum(res) == 0 else "DIFFER FROM"))
    print("rccl_loopback_probe: %d mismatches" % bad)
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()

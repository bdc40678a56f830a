#!/usr/bin/env python3
"""Can RCCL run several ranks of the sharded map on ONE GPU (a one-GPU box is all there is before the driver's scaling
run)?  Spawns W rank processes (default 2) that all use device 0: ShardedGvom + RcclComm over a few steps, every rank
compares its returned maps with an unsharded mapper fed the concatenated shares.  The parent makes no HIP call.
Prints what RCCL says; exit code 0 only if every rank ran and matched.  usage: tools/rccl_ranks_one_gpu.py [W]"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def rank_main(rank, world, name):
    for p in (ROOT, os.path.join(ROOT, "g-vom_amd")):
        sys.path.insert(0, p)
    import io, contextlib
    import numpy as np
    import gvom, gvom_sharded, synth
    params = (0.2, 0.2, 128, 32, 2, 1.0, 0.5, 0.5, 0.3, 2.0, 4.0, 1.0, 1, 1)
    comm = gvom_sharded.RcclComm(rank, world, 0, name)
    sh = gvom_sharded.ShardedGvom(*params, comm=comm, device=0)
    ref = gvom.Gvom(*params)
    scene = synth.make_scene(2, extent=11.0)
    for k in range(4):
        ego = (0.6 * k, -0.4 * k, 0.03 * k)
        shares = [synth.lidar_scan(scene, beams=16, azimuths=1024, sensor=ego, yaw=0.001 * r, noise_seed=10 * k + r)[:16384 - 991 * r]
                  for r in range(world)]
        sh.process_pointcloud(shares[rank], ego)
        with contextlib.redirect_stdout(io.StringIO()):
            ref.process_pointcloud(np.concatenate(shares, 0), ego)
        got, want = sh.combine_maps(), ref.combine_maps()
        for a, b in zip(got, want):
            assert np.array_equal(a, b), "rank %d step %d" % (rank, k)
        assert sh.combined_cell_count_cpu == ref.combined_cell_count_cpu
    comm.close()
    print("rank %d of %d on device 0: 4 steps, maps equal the unsharded mapper's" % (rank, world), flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "--rank":
        rank_main(int(sys.argv[2]), int(sys.argv[3]), sys.argv[4])
        sys.exit(0)
    W = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    name = "gvom_onegpu_%d" % os.getpid()
    env = dict(os.environ, GVOM_COMM_TIMEOUT_S="40", HSA_ENABLE_IPC_MODE_LEGACY="0")
    procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__), "--rank", str(r), str(W), name], env=env,
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT) for r in range(W)]
    rcs = []
    for r, p in enumerate(procs):
        try:
            out, _ = p.communicate(timeout=90)
        except subprocess.TimeoutExpired:
            p.kill()
            out, _ = p.communicate()
            out += b"\n[killed after 90 s]"
        rcs.append(p.returncode)
        print("---- rank %d (exit %s)\n%s" % (r, p.returncode, out.decode(errors="replace")[-1500:]))
    sys.exit(0 if all(rc == 0 for rc in rcs) else 1)

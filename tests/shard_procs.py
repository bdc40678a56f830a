#!/usr/bin/env python3
"""The sharded map with its ranks as PROCESSES on one GPU, through the library's own communicator (gvom_comm_*) and its
peer-copy transport (exported send regions pulled with hipMemcpyAsync; RCCL refuses two ranks on one device, this
transport does not): the multi-process side of the product -- shared-memory rendezvous, count exchange, exported
allocations re-exported when a region has grown, the scan's sparse all-to-all, the statistics exchange, the combine's
all-gather -- moving real device data between real processes.  Every rank compares what it gets with an unsharded
mapper fed the concatenated shares.  The launcher makes no HIP call.

usage: tests/shard_procs.py <world> [transport=peer] [stats=0] [repeat=1]      exit code 0 iff every rank ran and matched
(repeat > 1: the seven steps over and over on the same maps -- a soak of the transport)"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

PARAMS = (0.2, 0.2, 96, 32, 3, 1.0, 0.5, 0.5, 0.3, 2.0, 4.0, 1.0, 1, 1)


def plan(world):
    """the same steps on every rank: (shares per rank, ego, transform) -- ragged shares growing from scan to scan (the
    exchange regions are re-allocated and exported again), an empty share, a scan no rank accepts, a float64 scan
    with a transform"""
    import numpy as np
    sys.path.insert(0, os.path.join(ROOT, "g-vom_amd"))
    import synth
    scene = synth.make_scene(2, extent=8.0)
    steps = []
    for k in range(7):
        ego = (0.45 * k, -0.3 * k, 0.02 * k)
        n_az = 256 << min(k, 3)                                         # 4 k ... 32 k returns per rank
        shares = [synth.lidar_scan(scene, beams=16, azimuths=n_az, sensor=ego, yaw=0.0013 * r, noise_seed=10 * k + r)
                  for r in range(world)]
        shares = [s[:s.shape[0] - 197 * r] for r, s in enumerate(shares)]
        tf = None
        if k == 2:
            shares[world - 1] = shares[world - 1][:0]                   # a rank without returns
        if k == 4:
            shares = [s + np.float32(500.0) for s in shares]            # nothing lands in the grid: every rank rejects
        if k == 5:                                                      # sensor-frame float64 cloud + 4 x 4 transform
            tf = synth.sensor_transform(ego)
            ti = np.linalg.inv(tf)
            shares = [(s.astype(np.float64) @ ti[:3, :3].T + ti[:3, 3]) for s in shares]
        steps.append((shares, ego, tf))
    return steps


def rank_main(rank, world, name, transport, stats, repeat=1):
    for p in (ROOT, os.path.join(ROOT, "g-vom_amd")):
        sys.path.insert(0, p)
    import contextlib
    import io
    import numpy as np
    import gvom
    import gvom_sharded
    comm = gvom_sharded.RcclComm(rank, world, 0, name, transport=transport)
    assert comm.transport == ("peer" if transport in ("peer", "auto") else "rccl"), comm.transport
    sh = gvom_sharded.ShardedGvom(*PARAMS, comm=comm, device=0, voxel_statistics=stats)
    if os.environ.get("GVOM_TEST_CHURN"):                # every scan exports a fresh allocation (a re-export and a re-open per peer)
        sh.b.g.set_tuning("churn", int(os.environ["GVOM_TEST_CHURN"]))
    ref = gvom.Gvom(*PARAMS, voxel_statistics=stats)
    n_maps = 0
    steps = plan(world)
    for k, (shares, ego, tf) in enumerate(steps * repeat):
        with contextlib.redirect_stdout(io.StringIO()):
            sh.process_pointcloud(shares[rank], ego, tf)
            ref.process_pointcloud(np.concatenate(shares, 0), ego, tf)
            got, want = sh.combine_maps(), ref.combine_maps()
        assert (got is None) == (want is None), "rank %d step %d: combine presence" % (rank, k)
        if got is None:
            continue
        n_maps += 1
        for a, b in zip(got, want):
            assert a.dtype == b.dtype and np.array_equal(a, b, equal_nan=True), "rank %d step %d: returned maps" % (rank, k)
        assert sh.combined_cell_count_cpu == ref.combined_cell_count_cpu, "rank %d step %d: cell count" % (rank, k)
        if stats:
            # this rank's voxels of the debug voxel cloud; the ranks' rows together are the unsharded mapper's
            mine = sh.make_debug_voxel_map()
            counts = comm.exchange_host([0 if mine is None else int(mine.shape[0])])
            whole = ref.make_debug_voxel_map()
            assert sum(c[0] for c in counts) == (0 if whole is None else whole.shape[0]), "rank %d step %d: voxel count" % (rank, k)
            if mine is not None and mine.shape[0]:
                where = {whole[i, :3].tobytes(): i for i in range(whole.shape[0])}
                idx = np.array([where[mine[i, :3].tobytes()] for i in range(mine.shape[0])])
                assert np.array_equal(whole[idx][:, :5], mine[:, :5]), "rank %d step %d: voxel positions / counts" % (rank, k)
                np.testing.assert_allclose(mine[:, 5:], whole[idx][:, 5:], rtol=1e-4, atol=2e-5)
    comm.barrier()
    ps = comm.peer_stats()
    ps["open_fds"] = len(os.listdir("/proc/self/fd"))
    comm.close()
    print("rank %d of %d (%s transport%s): %d combines equal the unsharded mapper's; %s" %
          (rank, world, comm.transport, ", statistics" if stats else "", n_maps, ps), flush=True)


def launch(world, transport="peer", stats=False, timeout=240, repeat=1, asynchronous=False):
    """-> (ok, text).  Starts the rank processes and collects what they say."""
    name = "gvom_procs_%d_%d" % (os.getpid(), world)
    env = dict(os.environ, GVOM_COMM_TIMEOUT_S="120", HSA_ENABLE_IPC_MODE_LEGACY="0", GVOM_PEER_ASYNC="1" if asynchronous else "0")
    procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__), "--rank", str(r), str(world), name, transport,
                               "1" if stats else "0", str(repeat)], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
             for r in range(world)]
    ok, text = True, []
    for r, p in enumerate(procs):
        try:
            out, _ = p.communicate(timeout=timeout)
        except subprocess.TimeoutExpired:
            p.kill()
            out, _ = p.communicate()
            out += b"\n[killed after %d s]" % timeout
        ok = ok and p.returncode == 0
        text.append("---- rank %d (exit %s)\n%s" % (r, p.returncode, out.decode(errors="replace")[-1500:]))
    return ok, "\n".join(text)


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "--rank":
        rank_main(int(sys.argv[2]), int(sys.argv[3]), sys.argv[4], sys.argv[5], sys.argv[6] == "1", int(sys.argv[7]) if len(sys.argv) > 7 else 1)
        sys.exit(0)
    W = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    rep = int(sys.argv[4]) if len(sys.argv) > 4 else 1
    ok, text = launch(W, sys.argv[2] if len(sys.argv) > 2 else "peer", len(sys.argv) > 3 and sys.argv[3] == "1", timeout=240 + 2 * rep, repeat=rep)
    print(text)
    sys.exit(0 if ok else 1)

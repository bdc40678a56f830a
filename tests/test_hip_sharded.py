"""GPU tests of the sharded map (g-vom_amd/gvom_sharded.py, include/gvom_hip.h gvom_shard_* /
gvom_comm_*): the ranks run as threads of one process on the test box's one GPU
(tests/shard_threads.py) -- real pack / unpack / slab kernels and split C-ABI entry points, the SPMD
orchestration of the product, only the wire is hipMemcpyAsync on the receiving handle's stream instead of
ncclRecv / ncclAllGather on it (same ordering: tools/repro_shard_race.py).  Every rank's rows of every
ring slot and of the fused map, and every rank's returned maps, must equal the unsharded handle's,
bit for bit.  The RCCL binding itself is exercised with one rank."""
import io
import contextlib
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TEST_LIB = os.path.join(ROOT, "g-vom_amd", "lib", "libgvom_hip_test.so")     # production sources + the three test hooks


@pytest.mark.parametrize("world,transport", [(2, None), (4, None), (2, "loopback"), (4, "loopback")])
def test_sharded_map_equals_single_handle_over_a_moving_window(world, transport):
    """transport None: the thread double (device copies); "loopback": the product's communicator over RCCL -- every rank a
    1-rank communicator, grouped ncclSend / ncclRecv to itself on the receiving handle's stream in front of k_unpack_*, the
    in-place ncclAllGather in front of k_map2d, the library's one-call scan and combine (csrc/gvom_comm.hip)."""
    import gvom
    import synth
    from shard_threads import run_ranks
    params = (0.2, 0.2, 128, 32, 3, 1.0, 0.5, 0.5, 0.3, 2.0, 4.0, 1.0, 1, 1)
    scene = synth.make_scene(2, extent=11.0)
    steps = []
    for k in range(5):
        ego = (0.7 * k, -0.45 * k, 0.05 * k)
        # every rank its own sensor, with a different number of returns (ragged shares)
        shares = [synth.lidar_scan(scene, beams=16, azimuths=1024, sensor=ego, yaw=0.001 * r, noise_seed=10 * k + r)[:16384 - 997 * r]
                  for r in range(world)]
        if k == 3:
            shares = [s + 900.0 for s in shares]                 # globally rejected scan
        steps.append((shares, ego))
    ref = gvom.Gvom(*params)
    want = []
    with contextlib.redirect_stdout(io.StringIO()):
        for shares, ego in steps:
            ref.process_pointcloud(np.concatenate(shares, 0), ego)
            want.append((ref.combine_maps(), ref.combined_cell_count_cpu, ref.buffer_index))

    def body(r, sh):
        assert sh.combine_maps() is None
        for (shares, ego), (wout, wcnt, wbuf) in zip(steps, want):
            sh.process_pointcloud(shares[r], ego)
            got = sh.combine_maps()
            for a, b in zip(got, wout):
                assert a.dtype == b.dtype and np.array_equal(a, b)
            assert sh.combined_cell_count_cpu == wcnt
            assert sh.b.g.buffer_index == wbuf
        if transport == "loopback":
            assert sh.comm.transport == "loopback" and sh.comm.info()["rccl_comm_count"] == 1
            w = sh.comm.wire_stats()
            # every scan's exchange is one group (the rejected scan's too), every combine one more and one all-gather; a combine
            # moves every other rank's rows with one ncclSend + one ncclRecv, an accepted scan at least its quads or endpoints
            assert w["groups"] == 2 * len(steps) and w["allgathers"] == len(steps), w
            assert w["p2p_calls"] >= 2 * (world - 1) * (len(steps) + 4) and w["p2p_bytes"] > (1 << 20), w
        return True

    with contextlib.redirect_stdout(io.StringIO()):
        assert run_ranks(world, params, body, transport=transport) == [True] * world


@pytest.mark.parametrize("world,dtype,transport", [(2, np.float32, None), (4, np.float32, None), (4, np.float64, None),
                                                   (4, np.float64, "loopback")])
def test_sharded_voxel_statistics_equal_the_unsharded_handle(world, dtype, transport):
    """SURVEY 8f rank 2 on a sharded map ("absent on sharded handles"): with voxel_statistics=True every rank
    also receives the returns whose 27-voxel neighbourhood reaches into its rows, and make_debug_voxel_map() returns the
    rank's own voxels.  The ranks' rows together must be the unsharded mapper's debug cloud -- positions, hit counts and
    solid factors exactly, eigenvalue columns to the tolerance of the other statistics tests (float accumulation order is
    unspecified on both sides) -- over a moving window with ragged shares, and the returned maps stay bit-identical."""
    import gvom
    import synth
    from shard_threads import run_ranks
    params = (0.2, 0.2, 64, 32, 2, 1.0, 0.5, 0.5, 0.3, 2.0, 4.0, 1.0, 1, 1)
    scene = synth.make_scene(2, extent=5.5)
    steps = []
    for k in range(4):
        ego = (0.5 * k, -0.35 * k, 0.04 * k)
        shares = [synth.lidar_scan(scene, beams=16, azimuths=512, sensor=ego, yaw=0.002 * r, noise_seed=10 * k + r, dtype=dtype)[:8192 - 701 * r]
                  for r in range(world)]
        if k == 1:
            shares[1] = shares[1][:0]                            # a rank without returns in this scan
        steps.append((shares, ego))
    steps.insert(3, ([sh + dtype(900.0) for sh in steps[2][0]], steps[2][1]))       # a scan every rank rejects (gvom.py:147-150)
    ref = gvom.Gvom(*params, voxel_statistics=True)
    want = []
    for shares, ego in steps:
        ref.process_pointcloud(np.concatenate(shares, 0), ego)
        maps = ref.combine_maps()
        want.append((maps, ref.make_debug_voxel_map()))
    assert want[-1][1] is not None and want[-1][1].shape[0] > 500

    def body(r, sh):
        parts = []
        for (shares, ego), (wmaps, _) in zip(steps, want):
            sh.process_pointcloud(shares[r], ego)
            got = sh.combine_maps()
            for a, b in zip(got, wmaps):
                assert np.array_equal(a, b)
            parts.append(np.array(sh.make_debug_voxel_map(), copy=True))
        return parts

    parts = run_ranks(world, params, body, transport=transport, voxel_statistics=True)
    for k, (_, wcloud) in enumerate(want):
        got = np.concatenate([parts[r][k] for r in range(world)], 0)
        assert got.shape == wcloud.shape, (k, got.shape, wcloud.shape)
        g = got[np.lexsort((got[:, 2], got[:, 1], got[:, 0]))]
        w = wcloud[np.lexsort((wcloud[:, 2], wcloud[:, 1], wcloud[:, 0]))]
        assert np.array_equal(g[:, :5], w[:, :5]), "step %d: positions / solid factor / hit count" % k
        np.testing.assert_allclose(g[:, 5:], w[:, 5:], rtol=1e-4, atol=2e-5, err_msg="step %d: eigenvalue columns" % k)


@pytest.mark.parametrize("cfg,worlds,scans,buffer,transport", [("c2", "4,8", "3", "1", "threads"), ("c4", "4", "6", "4", "threads"),
                                                               ("c4", "2,4", "4", "4", "loopback")])
def test_sharded_map_at_full_size(cfg, worlds, scans, buffer, transport):
    """tests/fuzz/shard_big.py: the weak-scaling clouds of the bench on the c2 grid (4 and 8 ranks x
    131,072 returns) and BASELINE c4 at its own settings (512 x 512 x 128, buffer 4, 4 ranks x 262,144 returns, six
    scans of a moving window: the ring wraps and evicts): slots, fused map and returned maps of every rank equal the
    unsharded handle's.  "loopback": c4's clouds through the product's communicator over RCCL (2 and 4 ranks: 262,144 /
    524,288 returns per rank, quads by the ten thousand per exchange)."""
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "fuzz", "shard_big.py"), cfg, worlds, scans, buffer, transport],
                         capture_output=True, text=True, timeout=1200)
    assert out.returncode == 0, (out.stdout + out.stderr)[-2000:]
    assert "shard_big: 0 mismatches" in out.stdout


def test_a_rank_that_gives_up_ends_the_other_ranks_waits_at_once(monkeypatch):
    """gvom_comm_abort (RcclComm.abort): a rank whose CALLER fails outside the library -- here rank 1's body raises between two
    steps -- marks the communicator broken; rank 0, already waiting for it inside the next scan's count exchange, gets an error
    naming the rank within milliseconds instead of after GVOM_COMM_TIMEOUT_S.  Thread-ranks over the RCCL loopback transport
    (tests/shard_threads.run_ranks aborts for a failed body)."""
    import time
    import synth
    from shard_threads import run_ranks
    monkeypatch.setenv("GVOM_COMM_TIMEOUT_S", "120")
    params = (0.2, 0.2, 64, 32, 2, 1.0, 0.5, 0.5, 0.3, 2.0, 4.0, 1.0, 1, 1)
    scene = synth.make_scene(2, extent=5.5)
    share = synth.lidar_scan(scene, beams=16, azimuths=512, sensor=(0.0, 0.0, 0.0))
    seen = {}

    def body(r, sh):
        sh.process_pointcloud(share, (0.0, 0.0, 0.0))
        assert sh.combine_maps() is not None
        if r == 1:
            raise RuntimeError("rank 1 gives up")
        t0 = time.perf_counter()
        try:
            sh.process_pointcloud(share, (0.1, 0.0, 0.0))        # collective: rank 1 never comes
        finally:
            seen["waited_s"] = time.perf_counter() - t0
        return True

    with pytest.raises(Exception) as err:
        with contextlib.redirect_stdout(io.StringIO()):
            run_ranks(2, params, body, transport="loopback")
    assert "rank 1 gives up" in str(err.value)                   # the first failure is the one reported
    assert seen["waited_s"] < 10.0, seen


def test_sharded_product_path_over_rccl_with_one_rank():
    """The product path end to end -- ShardedGvom + RcclComm (ncclCommInitRank, the shared-memory count
    exchange, grouped send/recv with no peers, in-place ncclAllGather) -- with the one rank a one-GPU box
    allows; results equal the plain handle's."""
    import gvom
    import gvom_sharded
    import synth
    params, scans = synth.config_inputs("c2", n_scans=3)
    comm = gvom_sharded.RcclComm(0, 1, 0, "gvom_test1_%d" % os.getpid())
    try:
        sh = gvom_sharded.ShardedGvom(*params, comm=comm, device=0)
        ref = gvom.Gvom(*params)
        for pc, ego, tf in scans:
            sh.process_pointcloud(pc, ego, tf)
            ref.process_pointcloud(pc, ego, tf)
            for a, b in zip(sh.combine_maps(), ref.combine_maps()):
                assert a.dtype == b.dtype and np.array_equal(a, b)
            assert sh.combined_cell_count_cpu == ref.combined_cell_count_cpu
    finally:
        comm.close()


def test_rccl_binding_with_one_rank():
    """librccl.so is loaded by libgvom_hip.so itself (no PyTorch): communicator creation through the
    shared-memory rendezvous, the host-side exchange and an in-place all-gather of the library's own
    height-map buffer on the library's stream, with world = 1 (the test box has one GPU)."""
    import ctypes
    import gvom
    import gvom_sharded
    import synth
    comm = gvom_sharded.RcclComm(0, 1, 0, "gvom_test_%d" % os.getpid())
    try:
        assert comm.exchange_host([7, -3, 1 << 40]) == [[7, -3, 1 << 40]]
        comm.barrier()
        params, scans = synth.config_inputs("c2")
        g = gvom.Gvom(*params)
        pc, ego, tf = scans[0]
        g.process_pointcloud(pc, ego, tf)
        out = g.combine_maps()
        before = g._map2d(gvom.MAP_HEIGHT)

        class B(object):
            h = g._h
        comm.allgather_rows(B)
        g._check(g._lib.gvom_sync(g._h))
        assert np.array_equal(g._map2d(gvom.MAP_HEIGHT), before)
        assert out is not None
    finally:
        comm.close()


@pytest.mark.parametrize("world,stats", [(2, False), (4, False), (2, True)])
def test_rank_processes_on_one_gpu_through_the_peer_transport(world, stats):
    """tests/shard_procs.py: the ranks as PROCESSES, the library's own communicator (shared-memory rendezvous and count
    exchange) and its peer-copy transport (exported send regions, pulled by the receiver on its handle's stream) --
    RCCL refuses two ranks on one device, so this is the multi-process device exchange a one-GPU box can run: seven
    steps with growing ragged shares (regions re-allocated and exported again), an empty share, a scan every rank
    rejects and a float64 scan with a transform; every rank's maps and cell counts equal an unsharded mapper's, with
    statistics also the ranks' voxel clouds."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import shard_procs
    ok, text = shard_procs.launch(world, "peer", stats)
    assert ok, text[-3000:]
    assert text.count("combines equal the unsharded mapper's") == world, text[-3000:]


def test_rank_processes_through_the_asynchronous_peer_transport():
    """GVOM_PEER_ASYNC=1: the peer transport without host waits inside an exchange -- the copies are enqueued, one-thread kernels
    behind them write exchange numbers into the (HIP-registered) shared-memory segment, and a rank waits for its peers'
    numbers only before it overwrites what they pull from (gvom_comm_before_scan / _before_combine).  Four rank processes,
    the seven steps three times over, with the statistics exchange on two: every rank's results equal the unsharded mapper's."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import shard_procs
    ok, text = shard_procs.launch(4, "peer", False, repeat=3, asynchronous=True)
    assert ok, text[-3000:]
    assert text.count("'asynchronous': True") == 4, text[-3000:]
    ok, text = shard_procs.launch(2, "peer", True, asynchronous=True)
    assert ok, text[-3000:]
    assert text.count("'asynchronous': True") == 2, text[-3000:]


def test_peer_transport_with_a_new_exported_region_every_scan(monkeypatch):
    """The rules that keep inter-process memory honest, under the abuse that found them (test hook gvom_set_tuning("churn"): the
    endpoint send region is re-allocated, and therefore exported and mapped, on EVERY scan): an allocation exported once, a
    mapping opened once, regions of whole 2 MiB, nothing a peer may have mapped given back to the allocator.  Without any one of
    them this run ends in differing maps or a refused hipIpc call (profiles/r3_peer_churn.txt); four rank processes, 105 scans."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import shard_procs
    monkeypatch.setenv("GVOM_HIP_LIBRARY", TEST_LIB)           # ("churn" is a test hook: include/gvom_hip_test.h)
    monkeypatch.setenv("GVOM_TEST_CHURN", "1")
    ok, text = shard_procs.launch(4, "peer", False, repeat=15)
    assert ok, text[-3000:]
    assert text.count("105 combines equal the unsharded mapper's") == 4, text[-3000:]


@pytest.mark.parametrize("fault,world,asynchronous", [("export:2,rank:1", 2, False), ("import:3,rank:0", 2, False), ("export:1", 4, False),
                                                      ("import:2", 4, False), ("import:5,rank:2", 4, False), ("import:7,rank:3", 4, False),
                                                      ("import:2", 4, True), ("export:2,rank:1", 2, True), ("import:6,rank:1", 4, True)])
def test_peer_transport_absorbs_a_refused_export_or_import(monkeypatch, fault, world, asynchronous):
    """hipIpcGetMemHandle / hipIpcOpenMemHandle can refuse an allocation (profiles/r3_peer_churn.txt).  The library, not the
    test harness, absorbs it: the refused region moves into a fresh allocation which is exported instead (a refused OPEN is
    reported through the segment, its owner does the same, every rank tries again -- collectively, inside the exchange).
    Test hook GVOM_TEST_IPC_REFUSE: the N-th export / import of a process (of one rank, or of every rank) is answered with the
    runtime's refusal.  The run must end with the unsharded mapper's maps, no restart, and say how many regions it renewed.
    Late refusals (the 5th / 7th import of one rank: an export that has been on the table for several exchanges, opened for the
    first time by this rank -- ADVICE r4: whether a recovery round is needed is decided collectively, not from the export
    generations a rank has seen) and the asynchronous form of the transport (GVOM_PEER_ASYNC=1) take the same path."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import shard_procs
    monkeypatch.setenv("GVOM_HIP_LIBRARY", TEST_LIB)           # (the fault injector is a test hook: include/gvom_hip_test.h)
    monkeypatch.setenv("GVOM_TEST_IPC_REFUSE", fault)
    ok, text = shard_procs.launch(world, "peer", False, asynchronous=asynchronous)
    assert ok, text[-3000:]
    assert text.count("combines equal the unsharded mapper's") == world, text[-3000:]
    import re
    renewed = [int(m) for m in re.findall(r"'renewed_regions': (\d+)", text)]
    assert len(renewed) == world and sum(renewed) >= 1, text[-3000:]


def test_production_library_ignores_the_fault_injector(monkeypatch):
    """GVOM_TEST_IPC_REFUSE is read by lib/libgvom_hip_test.so only: under the production library the same environment renews
    nothing (no refusal is injected) and the run is an ordinary one."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import re
    import shard_procs
    monkeypatch.setenv("GVOM_TEST_IPC_REFUSE", "export:1")
    ok, text = shard_procs.launch(2, "peer", False)
    assert ok, text[-3000:]
    renewed = [int(m) for m in re.findall(r"'renewed_regions': (\d+)", text)]
    assert len(renewed) == 2 and sum(renewed) == 0, text[-3000:]


def test_auto_transport_falls_back_to_peer_copies_when_rccl_cannot_start():
    """GVOM_TRANSPORT_AUTO on one GPU with two rank processes: ncclCommInitRank refuses the duplicate device on both
    ranks, both agree on peer copies through the rendezvous, and the maps equal the unsharded mapper's."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import shard_procs
    ok, text = shard_procs.launch(2, "auto", False)
    assert ok, text[-3000:]
    assert text.count("(peer transport): ") == 2, text[-3000:]


# (the randomised campaign runs LAST: the deterministic full-size and RCCL tests above must not sit behind it
# under `pytest -x`)
def test_sharded_kernels_fuzz_against_unsharded_handle():
    """tests/fuzz/fuzz_shard.py on 90 edge-case seeds: 2 / 4 / 8 ranks, ragged and empty shares, tiny
    grids, rejected scans."""
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "fuzz", "fuzz_shard.py"), "70000", "90"],
                         capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    assert "checked 90 seeds, 0 failures" in out.stdout, out.stdout[-2000:]

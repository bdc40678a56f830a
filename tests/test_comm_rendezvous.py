"""The multi-process side of the sharded map without a GPU: ranks are PROCESSES that find each other
through the shared-memory rendezvous of libgvom_hip.so (gvom_comm_create with device -1 = no RCCL, no
HIP call) and run the per-scan host exchange (gvom_comm_exchange_host: the counts every rank needs
before the grouped ncclSend / ncclRecv).  What `bench.py --gpus N` relies on before its first
collective."""
import os
import struct
import subprocess
import sys
import time

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
WORKER = os.path.join(HERE, "_comm_worker.py")


def _run(world, name, rounds, delay_ms=0, die_at=-1):
    procs = [subprocess.Popen([sys.executable, WORKER, str(r), str(world), name, str(rounds), str(delay_ms), str(die_at)],
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT) for r in range(world)]
    outs = []
    for p in procs:
        try:
            out, _ = p.communicate(timeout=120)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            pytest.fail("rendezvous timed out")
        outs.append((p.returncode, out.decode()))
    return outs


@pytest.mark.parametrize("world", [2, 4])
def test_ranks_rendezvous_and_exchange(world):
    name = "gvom_test_%d_%d" % (os.getpid(), world)
    outs = _run(world, name, 2000)
    for r, (rc, out) in enumerate(outs):
        assert rc == 0 and ("ok %d" % r) in out, (r, rc, out)
    assert not os.path.exists("/dev/shm/" + name)                      # the name is gone once everybody has attached


def test_a_rank_that_dies_ends_the_others_wait_at_once():
    """A rank process that disappears between two exchanges (a crash, an exception that ends the interpreter): the
    others' next exchange must end in an error naming it within a moment, not after GVOM_COMM_TIMEOUT_S (600 s) --
    the segment records every rank's pid and start time and a wait of more than a few milliseconds looks them up
    (a zombie counts as gone: its launcher may not have reaped it yet)."""
    name = "gvom_test_dies_%d" % os.getpid()
    t0 = time.time()
    outs = _run(3, name, 200, die_at=120)
    assert time.time() - t0 < 30.0
    assert outs[2][0] == 7
    for r in (0, 1):
        assert outs[r][0] == 5 and "round 120" in outs[r][1] and "rank 2's process is gone" in outs[r][1], outs[r]


def test_a_segment_left_behind_by_a_crashed_job_is_not_joined():
    """Ranks 1.. poll before rank 0 has created the segment, and a stale file of the same name (old
    creation time, wrong contents) is lying in /dev/shm: they must wait for rank 0's fresh one."""
    name = "gvom_test_stale_%d" % os.getpid()
    path = "/dev/shm/" + name
    with open(path, "wb") as f:                                         # magic + world + id_ready + attached + created_s (a day ago)
        f.write(struct.pack("<IIIId", 0x47564F4D, 3, 1, 0, time.time() - 86400.0))
        f.write(b"\0" * (1 << 20))
    try:
        outs = _run(3, name, 200, delay_ms=300)
        for r, (rc, out) in enumerate(outs):
            assert rc == 0 and ("ok %d" % r) in out, (r, rc, out)
    finally:
        if os.path.exists(path):
            os.unlink(path)


def test_a_fresh_segment_of_a_job_that_crashed_minutes_ago_is_not_joined():
    """a job that crashed half a minute ago leaves a complete, fresh-looking segment behind (magic,
    id ready, fewer ranks attached than the world, young): ranks that start before rank 0 must not take its
    stale ncclUniqueId.  The segment names its creator (pid + start time); a creator that is no longer alive
    disqualifies it, and the ranks wait for the one rank 0 of THIS run creates."""
    name = "gvom_test_crashed_%d" % os.getpid()
    path = "/dev/shm/" + name
    dead = subprocess.Popen([sys.executable, "-c", "pass"])
    dead.wait()
    with open(path, "wb") as f:      # magic, world, id_ready, attached (1 < 3), created_s (30 s ago), creator pid + start time
        f.write(struct.pack("<IIIIdqQ", 0x47564F4D, 3, 1, 1, time.time() - 30.0, dead.pid, 12345))
        f.write(b"\0" * (1 << 20))
    try:
        outs = _run(3, name, 200, delay_ms=400)
        for r, (rc, out) in enumerate(outs):
            assert rc == 0 and ("ok %d" % r) in out, (r, rc, out)
    finally:
        if os.path.exists(path):
            os.unlink(path)


def test_rank_zero_removes_its_segment_when_creation_fails():
    """a failed gvom_comm_create on rank 0 must not leave /dev/shm/<name> behind for a later run to join
    (here: the device does not exist, so hipSetDevice / the RCCL load fails after nothing or before the segment;
    either way no file may remain)"""
    import ctypes
    sys.path.insert(0, os.path.join(os.path.dirname(HERE), "g-vom_amd"))
    import gvom
    lib = gvom.load_library()
    name = "gvom_test_fail_%d" % os.getpid()
    c = ctypes.c_void_p()
    rc = lib.gvom_comm_create(0, 2, 4096, name.encode(), ctypes.byref(c))
    assert rc != 0 and not c
    assert not os.path.exists("/dev/shm/" + name)


def test_transport_selection_and_bookkeeping_of_a_host_only_communicator():
    """gvom_comm_create2's transport argument: unknown values are refused (library and binding), a host-only communicator
    (device -1) takes any of them without touching RCCL or HIP, reports a transport and all-zero peer-copy statistics."""
    import ctypes
    sys.path.insert(0, os.path.join(os.path.dirname(HERE), "g-vom_amd"))
    import gvom
    import gvom_sharded
    lib = gvom.load_library()
    c = ctypes.c_void_p()
    assert lib.gvom_comm_create2(0, 1, -1, b"gvom_test_tr_%d" % os.getpid(), 7, ctypes.byref(c)) != 0 and not c.value
    with pytest.raises(KeyError):
        gvom_sharded.RcclComm(0, 1, -1, "gvom_test_tr2_%d" % os.getpid(), transport="carrier pigeon")
    for tr, want in (("rccl", "rccl"), ("peer", "peer"), ("auto", "rccl")):
        comm = gvom_sharded.RcclComm(0, 1, -1, "gvom_test_tr3_%d_%s" % (os.getpid(), tr), transport=tr)
        try:
            assert comm.transport == want
            assert comm.peer_stats() == {"bytes": 0, "copies": 0, "exports": 0, "open_retries": 0, "asynchronous": False, "renewed_regions": 0}
            info = comm.info()                      # a host-only communicator: no RCCL, no device, no bus id
            assert info["rccl_comm_count"] is None and info["rccl_user_rank"] is None and info["pci_bus_id"] == ""
            comm.before_scan(); comm.before_combine()                       # no-ops here
            assert comm.exchange_host([3, 4]) == [[3, 4]]
        finally:
            comm.close()

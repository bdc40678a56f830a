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


def _run(world, name, rounds, delay_ms=0):
    procs = [subprocess.Popen([sys.executable, WORKER, str(r), str(world), name, str(rounds), str(delay_ms)],
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

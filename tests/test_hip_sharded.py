"""GPU test of the slab-sharded HIP path: 2 and 4 ranks share the one GPU of the test box (the
collectives run over gloo, staged through the host), so the slab filters of k_trace / k_encode
/ k_fuse / k_map2d and the split C-ABI entry points run on real hardware; the result must be
bit-identical to the unsharded handle."""
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


@pytest.mark.parametrize("world", [2, 4])
def test_sharded_hip_equals_single_handle(world):
    port = _free_port()
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), LOCAL_RANK="0",
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), OMP_NUM_THREADS="1")
        procs.append(subprocess.Popen([sys.executable, os.path.join(HERE, "_sharded_hip_worker.py")], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    for r, p in enumerate(procs):
        try:
            out, _ = p.communicate(timeout=900)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        assert p.returncode == 0, "rank %d failed:\n%s" % (r, out.decode()[-3000:])


@pytest.mark.gpu
def test_sharded_kernels_fuzz_against_unsharded_handle():
    """tests/fuzz/fuzz_shard.py on 90 edge-case seeds: 2/4/8 sharded handles on one GPU, every rank's rows of
    every ring slot and of the fused map equal the unsharded handle's (scan culling, per-segment
    culling, slab encode and slab fusion)."""
    import subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "tests", "fuzz", "fuzz_shard.py"), "70000", "90"],
                         capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    assert "checked 90 seeds, 0 failures" in out.stdout, out.stdout[-2000:]


@pytest.mark.gpu
def test_sharded_kernels_on_weak_scaling_clouds():
    """tests/fuzz/shard_big.py: 4 and 8 sharded handles on one GPU fed the bench's weak-scaling clouds
    (524 k / 1 M returns: the 3-segment trace path of sharded handles); every rank's rows of the slot
    and of the fused map equal the unsharded handle's over three moving scans."""
    import subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "tests", "fuzz", "shard_big.py"), "4,8"],
                         capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, (out.stdout + out.stderr)[-2000:]
    assert "shard_big: 0 mismatches" in out.stdout

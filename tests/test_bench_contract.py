"""CPU-side checks of bench.py's contract plumbing (no GPU): the self-spawning `--gpus N` parent starts N rank
processes with the launcher environment and relays failure loudly instead of hanging; the roofline object
has the keys the contract names."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def test_roofline_object_has_the_contract_keys():
    import bench
    stages = {s: {"median": m, "p10": m * 0.9, "p90": m * 1.1, "samples": 20}
              for s, m in (("trace", 0.040), ("encode", 0.018), ("fuse", 0.020), ("map2d", 0.032))}
    alg = {"trace": 42.1e6, "encode": 335e6, "fuse": 200e6, "map2d": 4.4e6}
    r = bench.roofline_of(alg, stages, profiled=False)
    for key in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert key in r
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["kernel"] == "k_trace"
    assert abs(r["achieved"] - 42.1e6 / 40e-6 / 1e9) < 1e-6 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12
    assert r["traffic"] is None and r["valu"] is None                 # counter fields only for the profiled workload
    json.dumps(r)
    rp = bench.roofline_of(alg, stages, profiled=True)                # committed m256 passes
    assert rp["traffic"] and rp["valu"] and rp["valu"]["insts_per_launch"] > 1e6


def test_self_spawned_ranks_fail_loudly_without_gpus():
    """`python bench.py --gpus 2` without a launcher: the parent (which makes no HIP call) starts two ranks with
    RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set; without a GPU they raise, and the parent reports the exit codes
    and exits non-zero -- it neither hangs nor prints a JSON line."""
    if os.path.exists("/dev/kfd"):
        pytest.skip("a GPU driver is present: the ranks would run")
    env = dict(os.environ, GVOM_COMM_TIMEOUT_S="20")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0",
                        "--no-cpu"], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    assert p.returncode != 0
    assert b"rank exit codes" in p.stderr
    assert not [ln for ln in p.stdout.decode().splitlines() if ln.startswith("{")]


@pytest.mark.gpu
def test_bench_prints_one_json_line_with_the_contract_keys():
    """`python bench.py` (short run): ONE JSON line on stdout with the keys the driver reads, the roofline of the
    dominant kernel measured live, and the CPU baseline of the same workload."""
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "40", "--warmup", "10", "--no-extra",
                        "--cpu-budget", "2"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert p.returncode == 0, p.stderr.decode()[-2000:]
    lines = [ln for ln in p.stdout.decode().splitlines() if ln.strip()]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in d, key
    assert d["n_gpus"] == 1 and d["steps"] == 40 and d["warmup"] == 10 and d["higher_is_better"] is True
    assert d["unit"] == "M points/s" and d["vs_baseline"] is None and d["data"] == "synthetic"
    assert "workload" in d["config"] and "model" not in d["config"]
    assert abs(d["value"] - d["config"]["points_per_scan"] / (d["ms_per_step"] * 1e-3) / 1e6) < 1e-6 * d["value"]
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and 0 < r["frac"] < 1 and r["peak"] == 8000.0
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] == 1 and c["value"] > 0 and "sample" in c
    assert d["value"] > 50 * c["value"]


@pytest.mark.gpu
def test_sharded_bench_leg_with_one_rank():
    """The N > 1 leg of bench.py (one process per GPU, RCCL bound by libgvom_hip.so, shared-memory rendezvous) run
    with ONE rank -- what a one-GPU box allows: the same code path the driver's scaling run takes, JSON contract
    included."""
    env = dict(os.environ, GVOM_BENCH_FORCE_SHARDED="1")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "40", "--warmup", "10",
                        "--no-cpu"], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert p.returncode == 0, p.stderr.decode()[-2000:]
    lines = [ln for ln in p.stdout.decode().splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 1 and d["scaling"] == "weak" and d["steps"] == 40 and d["value"] > 100
    assert "sharded" in d["config"]["workload"] and d["config"]["points_per_gpu"] == d["config"]["points_per_step"]

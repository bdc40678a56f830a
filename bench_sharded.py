"""N > 1 leg of bench.py: one rank per GPU, one G-VOM map sharded into world-anchored y-slabs
(g-vom_amd/gvom_sharded.py), RCCL over xGMI called from libgvom_hip.so -- no PyTorch.

Workloads (--config):
  m256 (default), c2, c3, m256b8   weak scaling: every rank contributes one OS1-64 / OS1-128-shaped scan per step
                                   (a rig of N sensors), so a step processes N x 131,072 (262,144) points into ONE
                                   shared map, followed by one combine_maps
  c4   BASELINE.json config 4: 512 x 512 x 128 grid, buffer 4, ONE 1,048,576-point cloud (4 interleaved OS1-128
       sensors) per step, split by sensor over the ranks (1, 2 or 4 ranks; 4 x 262,144 is the config as quoted)
  c5   BASELINE.json config 5: 1024 x 1024 x 128 grid, buffer 8, 4,194,304 points per tick (16 sensors), split by
       sensor over 1 / 2 / 4 / 8 / 16 ranks (8 x 524,288 is the config as quoted)
Any other combination is refused.

The line carries its own correctness verdict: after the timed region a FRESH sharded map and, on rank 0, a fresh
unsharded gvom.Gvom are driven through the same buffer + 2 steps (the unsharded one with the concatenated shares);
`sharded_equals_unsharded` says whether rank 0's returned maps and fused cell counts were identical at every step
and every rank returned the same maps (checksums).  false -> exit code 3."""
import os
import sys
import time
import zlib

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "g-vom_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np

# name -> (grid config of synth.CONFIGS, sensors of the whole cloud or None = one sensor per rank, allowed world sizes)
WORKLOADS = {
    "m256": ("m256", None, None), "c2": ("c2", None, None), "c3": ("c3", None, None), "m256b8": ("m256b8", None, None),
    "c4": ("c4", 4, (1, 2, 4)), "c5": ("c5", 16, (1, 2, 4, 8, 16)),
}


def workload(name, world):
    """(params, sensors per rank, description); raises ValueError for a combination that does not exist."""
    import synth
    if name not in WORKLOADS:
        raise ValueError("bench.py --gpus N: no sharded workload %r (have: %s)" % (name, ", ".join(sorted(WORKLOADS))))
    grid, sensors, worlds = WORKLOADS[name]
    params, beams, desc = synth.CONFIGS[grid]
    if worlds is not None and world not in worlds:
        raise ValueError("config %s splits its %d sensors over %s ranks, not %d" % (name, sensors, "/".join(map(str, worlds)), world))
    if params[2] % (4 * world):
        raise ValueError("config %s: xy_size %d is not a multiple of 4 x %d ranks" % (name, params[2], world))
    per_rank = 1 if sensors is None else sensors // world
    return params, beams, per_rank, desc


def make_share(name, rank, world, k):
    """this rank's share of the cloud of step k (host array) and the sensor pose.  The shares of all ranks,
    concatenated in rank order, are the cloud the unsharded mapper gets."""
    import synth
    grid, sensors, _ = WORKLOADS[name]
    params, beams, _ = synth.CONFIGS[grid]
    if sensors is None:                                   # one sensor per rank: same pose, azimuth comb shifted
        pose = (0.2 * k, 0.0, 0.0)
        scene = synth.make_scene(2)
        return synth.lidar_scan(scene, beams=beams, sensor=pose, yaw=2.0 * np.pi / 2048 * rank / world,
                                noise_seed=100 * k + rank), pose
    per_rank = sensors // world                           # the cloud of synth.config_inputs(name), by sensor
    pose = (0.2 * k, -0.1 * k, 0.0)
    scene = synth.make_scene(2, extent=0.2 * params[2] / 2 * 0.9)
    parts = [synth.lidar_scan(scene, beams=beams, sensor=pose, yaw=2 * np.pi / 2048 * s / sensors, noise_seed=100 * k + s)
             for s in range(rank * per_rank, (rank + 1) * per_rank)]
    return np.concatenate(parts, 0), pose


FELL_BACK = False        # set by run(): AUTO ended up on peer copies (a helper thread may still sit inside ncclCommInitRank)


class ExchangeTimer(object):
    """Wraps a communicator: HIP events on the handle's stream around the scan's exchange and the combine's
    all-gather (the collectives run on that stream), read back after the step has synchronised."""

    def __init__(self, comm, hip):
        self.c, self.hip = comm, hip
        self.rank, self.world = comm.rank, comm.world
        self.on = False
        self.ev = [hip.event_create() for _ in range(4)]
        self.ms = {"exchange_scan": [], "allgather_rows": []}

    def exchange_host(self, v):
        return self.c.exchange_host(v)

    def barrier(self):
        return self.c.barrier()

    def before_scan(self):
        return self.c.before_scan()

    def before_combine(self):
        return self.c.before_combine()

    def exchange_scan(self, backend, sq, se, rq, re):
        if not self.on:
            return self.c.exchange_scan(backend, sq, se, rq, re)
        st = backend.lib.gvom_stream(backend.h)
        self.hip.event_record(self.ev[0], st)
        self.c.exchange_scan(backend, sq, se, rq, re)
        self.hip.event_record(self.ev[1], st)

    def allgather_rows(self, backend):
        if not self.on:
            return self.c.allgather_rows(backend)
        st = backend.lib.gvom_stream(backend.h)
        self.hip.event_record(self.ev[2], st)
        self.c.allgather_rows(backend)
        self.hip.event_record(self.ev[3], st)

    def collect(self):
        self.ms["exchange_scan"].append(self.hip.event_elapsed_ms(self.ev[0], self.ev[1]))
        self.ms["allgather_rows"].append(self.hip.event_elapsed_ms(self.ev[2], self.ev[3]))


def _checksum(out):
    c = 0
    for a in out:
        c = zlib.crc32(np.ascontiguousarray(a).view(np.uint8).reshape(-1), c)
    return c


def verify(name, params, comm, rank, world, local_rank, n_steps):
    """A fresh sharded map against a fresh unsharded one (rank 0) over n_steps steps from an empty ring.
    Returns (equal, detail) on rank 0, (None, None) elsewhere.  Collective."""
    import contextlib
    import io
    import gvom
    import gvom_sharded
    sh = gvom_sharded.ShardedGvom(*params, comm=comm, device=local_rank)
    ref = gvom.Gvom(*params, device=local_rank) if rank == 0 else None
    differing, steps_bad, crc_bad, trace_alg = 0, 0, 0, None
    for k in range(n_steps):
        share, pose = make_share(name, rank, world, k)
        sh.process_pointcloud(share, pose)
        out = sh.combine_maps()
        cells = sh.combined_cell_count_cpu                              # collective
        crcs = comm.exchange_host([_checksum(out)])
        if rank != 0:
            continue
        if any(c[0] != crcs[0][0] for c in crcs):
            crc_bad += 1
        if k == 0:                                                     # exact accounting of THIS rank's k_trace launch
            with contextlib.redirect_stdout(io.StringIO()):
                solo = gvom.Gvom(*params, device=local_rank)
                solo.process_pointcloud(share, pose)
                st = solo.scan_stats()
            if st:
                trace_alg = {"points": int(share.shape[0]), "sum_hit": st["sum_hit"], "sum_total": st["sum_total"],
                             "bytes": int(share.shape[0]) * 12 + 4 * (st["sum_hit"] + st["sum_total"])}
            del solo
        cloud = np.concatenate([share] + [make_share(name, r, world, k)[0] for r in range(1, world)], 0)
        ref.process_pointcloud(cloud, pose)
        want = ref.combine_maps()
        bad = sum(int(np.count_nonzero(a != b)) for a, b in zip(out, want))
        bad += int(cells != ref.combined_cell_count_cpu)
        differing += bad
        steps_bad += 1 if bad else 0
    del sh
    if rank != 0:
        return None, None, None
    ok = differing == 0 and crc_bad == 0
    return ok, {"steps": n_steps, "differing_cells": differing, "steps_with_differences": steps_bad,
                "steps_where_ranks_disagree": crc_bad,
                "what": "fresh ShardedGvom vs fresh gvom.Gvom fed the concatenated shares: origin, positive, negative, "
                        "roughness, visibility and the fused cell count after every step, bit for bit; every rank's "
                        "maps equal rank 0's (crc32)"}, trace_alg


def run(args):
    import bench
    import gvom_sharded
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", str(rank)))
    params, beams, per_rank, desc = workload(args.config, world)       # refuses before anything touches the GPU
    hip = bench.Hip()
    ndev = hip.device_count()
    if ndev < 1:
        raise RuntimeError("rank %d sees no HIP device" % rank)
    if args.share_device:                             # rehearsal: all ranks on one GPU
        local_rank = 0
    if local_rank >= ndev:                            # a launcher that shows every rank only its own GPU
        if os.environ.get("HIP_VISIBLE_DEVICES") is None and os.environ.get("ROCR_VISIBLE_DEVICES") is None and ndev < world:
            raise RuntimeError("rank %d: %d ranks but only %d HIP device(s) visible" % (rank, world, ndev))
        local_rank = local_rank % ndev
    affinity = bench.pin_to_gpu_numa(local_rank)      # each rank next to its own GPU
    hip.set_device(local_rank)
    name = args.config
    poses = max(1, min(args.poses, 8 if per_rank == 1 else 4))
    scans = []
    for k in range(poses):
        pc, pose = make_share(name, rank, world, k)
        scans.append(((hip.to_device(pc).value, pc.shape[0], pc.dtype), pose))
    n_local = scans[0][0][1]
    rccl = gvom_sharded.RcclComm(rank, world, local_rank, gvom_sharded.rendezvous_name(),
                                 transport="peer" if (args.share_device and not os.environ.get("GVOM_BENCH_REHEARSE_AUTO")) else args.transport)
    comm = ExchangeTimer(rccl, hip)
    sh = gvom_sharded.ShardedGvom(*params, comm=comm, device=local_rank)
    big = name in ("c4", "c5")
    steps = min(args.steps, 40) if big else args.steps
    warmup = min(args.warmup, 12) if big else args.warmup

    def step(k):
        share, ego = scans[k % poses]
        sh.process_pointcloud(share, ego)
        return sh.combine_maps()

    def fence():                                     # device idle on every rank, then a barrier, both sides of the timed region
        sh.b.sync(); comm.barrier(); sh.b.sync()

    for k in range(warmup):
        step(k)
    import gc
    gc.collect(); gc.disable()
    blocks, total, k = [], 0.0, warmup
    # blocks of exactly `steps` steps until >= 0.5 s has been timed on rank 0's clock (the ranks agree on the count)
    while True:
        fence()
        t0 = time.perf_counter()
        for _ in range(steps):
            step(k); k += 1
        fence()
        dt = time.perf_counter() - t0
        dt = max(r[0] for r in comm.exchange_host([int(dt * 1e9)])) * 1e-9      # MAX over ranks
        blocks.append(dt); total += dt
        if (total >= bench.MIN_TIMED_S and len(blocks) >= 3) or len(blocks) >= 200:
            break
    gc.enable()
    med = sorted(blocks)[len(blocks) // 2]
    # this rank's kernels (HIP events on the library's stream) and its exchanges, outside the timed blocks
    comm.on = True
    wire = {"sent": [], "received": []}

    def sampled_step(kk):
        out = step(kk)
        comm.collect()
        wire["sent"].append(sh.last_exchange_bytes[0]); wire["received"].append(sh.last_exchange_bytes[1])
        return out
    stages = bench.stage_samples(sh.b.g, sampled_step, k, 20)
    comm.on = False
    stage_ms = {s: v["median"] for s, v in stages.items()}
    ex_ms = {s: bench._median(v) for s, v in comm.ms.items()}
    sent, recvd = bench._median(wire["sent"]), bench._median(wire["received"])
    # every rank's exchange figures on rank 0 (the slowest rank sets the step)
    table = comm.exchange_host([int(sent), int(recvd), int((ex_ms["exchange_scan"] or 0) * 1e6), int((ex_ms["allgather_rows"] or 0) * 1e6)])
    fence()
    # the communicator's own view of the job, every rank's on rank 0: RCCL's rank count and rank number, device, PCI bus id
    info = rccl.info()
    bus = info["pci_bus_id"] or "?"
    packed = [int.from_bytes(bus.encode()[:21].ljust(21, b"\0")[i:i + 7], "little") for i in (0, 7, 14)]    # (7 bytes per int64)
    rows_info = comm.exchange_host([info["rccl_comm_count"] if info["rccl_comm_count"] is not None else -1,
                                    info["rccl_user_rank"] if info["rccl_user_rank"] is not None else -1,
                                    info["device"], 0 if info["transport"] == "rccl" else 1] + packed)

    def _bus(*parts):
        return b"".join(int(v).to_bytes(7, "little") for v in parts).rstrip(b"\0").decode(errors="replace")
    stream = None
    if getattr(args, "offered_hz", 0) > 0:
        # the paced stream (bench.paced_stream): every rank hands its HOST-resident share over at the common schedule
        # (the ranks are processes of one node: one monotonic clock); rank 0 reports, with the slowest rank's percentiles
        host_shares = [make_share(name, rank, world, kk) for kk in range(poses)]

        def tick(kk):
            share, ego = host_shares[kk % poses]
            sh.process_pointcloud(share, ego)
            return sh.combine_maps()
        for kk in range(3):                                 # warm ticks FIRST, then the ranks agree on the schedule's start
            tick(kk)
        t0 = max(r[0] for r in comm.exchange_host([int((time.perf_counter() + 0.25) * 1e9)])) * 1e-9
        stream = bench.paced_stream(tick, args.offered_hz, args.ticks, t0=t0, warm=0)
        rank_rows = comm.exchange_host([int(stream["latency_ms"][q] * 1e3) for q in ("p50", "p95", "p99", "max")] + [stream["deadline_misses"]])
        stream["per_rank"] = [{"rank": r, "latency_us_p50_p95_p99_max": list(row[:4]), "deadline_misses": row[4]} for r, row in enumerate(rank_rows)]
        stream["latency_ms_slowest_rank"] = {q: max(row[i] for row in rank_rows) * 1e-3 for i, q in enumerate(("p50", "p95", "p99", "max"))}
        stream.update(points_per_tick=n_local * world, offered_M_points_s=n_local * world * args.offered_hz / 1e6,
                      sustained_M_points_s=n_local * world * stream["achieved_hz"] / 1e6)
        del host_shares
        fence()
    n_ver = params[4] + 2
    ok, detail, trace_alg = verify(name, params, comm, rank, world, local_rank, n_ver)
    out = None
    if rank == 0:
        n_total = n_local * world
        tr_ms = stage_ms.get("trace")
        achieved = (trace_alg["bytes"] / (tr_ms * 1e-3) / 1e9) if (trace_alg and tr_ms) else None
        worst = max(table, key=lambda r: r[2])
        out = {
            "metric": bench.metric_for(name, list(params[2:3]) * 2 + [params[3]]), "value": n_total * steps / med / 1e6, "unit": "M points/s",
            "n_gpus": world, "steps": steps, "warmup": warmup,
            "ms_per_step": med / steps * 1e3, "higher_is_better": True,
            "scaling": "weak" if per_rank == 1 and not big else "strong",
            "vs_baseline": None, "dtype": bench.DTYPE, "data": "synthetic",
            "config": {"workload": desc + ("; %d sensors, one shared map, grid sharded into %d y-slabs" % (world, world) if not big else
                                           "; cloud split by sensor over %d rank(s), grid sharded into %d y-slabs" % (world, world)),
                       "name": name, "points_per_step": n_total, "points_per_gpu": n_local,
                       "grid": [params[2], params[2], params[3]], "buffer_size": params[4], "poses": poses,
                       "input": "device-resident f32 xyz", "host_affinity_rank0": affinity,
                       "timing": "median of %d blocks of %d steps, max over ranks" % (len(blocks), steps),
                       "exchange": "per scan: sparse all-to-all of dirty accumulator quads (1 KiB + id) and endpoints (8 B); "
                                   "per combine: in-place all-gather of height|inferred|density rows (24 B/cell); counts "
                                   "through shared memory; transport: see \"transport\""},
            "transport": {"rccl": "RCCL (grouped ncclSend/ncclRecv, ncclAllGather)",
                          "peer": "peer copies (exported regions, hipMemcpyAsync pulled by the receiver; %s)" %
                                  ("asynchronous: completion flags written by the GPUs" if rccl.peer_async else "two host barriers and stream waits per exchange")}[rccl.transport]
                         + ("" if args.transport != "auto" or rccl.transport == "rccl" or (args.share_device and not os.environ.get("GVOM_BENCH_REHEARSE_AUTO"))
                            else " -- RCCL could not initialise"),
            "rehearsal_on_one_device": bool(args.share_device),
            "ranks": [{"rank": r, "rccl_comm_count": (row[0] if row[0] >= 0 else None), "rccl_user_rank": (row[1] if row[1] >= 0 else None),
                       "device": row[2], "transport": ("rccl", "peer")[row[3]], "pci_bus_id": _bus(row[4], row[5], row[6])}
                      for r, row in enumerate(rows_info)],
            "distinct_devices": len(set(_bus(row[4], row[5], row[6]) for row in rows_info)),
            "communicator_ranks": (rows_info[0][0] if rows_info[0][0] >= 0 else world),
            "cpu_baseline": {"see": "the N = 1 line of the same bench.py (cpu_baseline: the CPU oracle on this box's host cores, timed on "
                                    "rank 0 at N = 1 only, as the contract asks); BENCH_rNN.json / profiles/r4_bench_m256.json"},
            "peer_transport_rank0": rccl.peer_stats() if rccl.transport == "peer" else None,
            # what RCCL was actually asked to move by rank 0 since the communicator was made (ncclSend + ncclRecv calls, their bytes,
            # groups closed, ncclAllGather calls): non-zero on every N > 1 run over RCCL -- the data path has executed
            "rccl_calls_rank0": rccl.wire_stats(),
            "map_hz": steps / med, "blocks": len(blocks),
            "ms_per_step_min": min(blocks) / steps * 1e3, "ms_per_step_max": max(blocks) / steps * 1e3,
            "stage_ms_rank0": stage_ms, "stream": stream,
            "sharded_equals_unsharded": bool(ok), "verify": detail,
            "exchange": {"per_rank": [{"rank": r, "sent_bytes": row[0], "received_bytes": row[1],
                                       "exchange_scan_ms": row[2] * 1e-6, "allgather_rows_ms": row[3] * 1e-6}
                                      for r, row in enumerate(table)],
                         "slowest_rank_wire_GBs": (max(worst[0], worst[1]) / (worst[2] * 1e-9) / 1e9) if worst[2] else None,
                         "note": "HIP events on the handle's stream around the scan's exchange and the combine's all-gather "
                                 "(they include the wait for the slowest peer); bytes = quads x 1028 + endpoints x 8"},
            "roofline": {"bound": "hbm", "kernel": "k_trace", "achieved": achieved, "peak": 8000.0, "unit": "GB/s",
                         "frac": achieved / 8000.0 if achieved else None, "traffic": None,
                         "avg_launch_ms": tr_ms, "algorithmic_bytes_per_launch": trace_alg["bytes"] if trace_alg else None,
                         "accounting": trace_alg,
                         "note": "rank 0's launch: k_trace walks this rank's own rays over the whole window (the same launch "
                                 "as on one GPU); algorithmic bytes = N*12 + 4*(sum_hit + sum_total) of exactly these rays; "
                                 "PMC traffic is profiled on the N = 1 run"},
        }
    fence()
    global FELL_BACK
    FELL_BACK = args.transport == "auto" and rccl.transport == "peer" and not (args.share_device and not os.environ.get("GVOM_BENCH_REHEARSE_AUTO"))
    rccl.close()
    return out

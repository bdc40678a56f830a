"""gvom_sharded -- one G-VOM map sharded over the GPUs of one node (one process per GPU).

The reference has no multi-GPU path; semantics are SURVEY 8(e): rays are data-parallel,
the per-voxel accumulators (hit / total: int32 sum, min-height: f32 min) are reduced onto the rank
that owns the voxel, everything after is per voxel / per column on the owner.  The result equals
gvom.Gvom fed with the concatenated cloud, bit for bit.

Partitioning: the voxel grid is stored world-anchored (toroidal), so its storage y axis is cut
into `world` contiguous slabs that NEVER migrate when the robot-centred window moves.  Rank r owns
storage rows [r*xy/world, (r+1)*xy/world) of every ring slot and of the fused map.

Per scan (process_pointcloud; every rank passes ITS share, shares may differ in length or be empty):
    local      k_trace over the rank's own rays and the WHOLE window into private accumulators
               (balanced: every rank walks exactly its own rays); endpoints that fall into another
               rank's rows are listed for their owner instead of being added locally
    local      k_pack: the dirty quads (4 rows x 64 sx at one sz = 1 KiB of ray-pass counts) of
               every other rank's rows -> that rank's send region; counts -> host-mapped memory
    host       the per-destination counts and the "some return in the grid" flags are exchanged
               through shared memory (the ranks are the processes of one node): ~1 us, no GPU
    exchange   sparse all-to-all: grouped ncclSend / ncclRecv of {quad ids, quads, endpoints}
               (RCCL over xGMI, on the library's stream, sizes exact)
    local      k_unpack: total += received quads, endpoint work of the received endpoints;
               k_encode over the rank's own rows; commit iff ANY rank saw a return in the grid
Per combine (combine_maps):
    local      temporal fusion + column reductions of the slab -> its rows of the height and
               inferred-height maps, plus the positive-obstacle density of its cells
    exchange   in-place ncclAllGather of the library's [sy][height | inferred | density] rows
               (24 B x xy^2 = 1.5 MB at xy = 256): the slope stencil needs +-1 row, __guess_height +-15
    local      every rank computes ALL rows of slope / roughness / guess / positive / negative /
               visibility (20 us of redundant 2-D work instead of a second collective) straight
               into pinned host memory; every rank returns the maps

No PyTorch: the collectives are RCCL calls made by libgvom_hip.so itself (csrc/gvom_comm.hip), bound
here with ctypes.  The orchestration below is transport-agnostic: `RcclComm` is the product (its "loopback"
transport runs several ranks as threads of ONE process on ONE GPU over RCCL); tests/ holds the doubles
(shard_threads.ThreadComm: plain device copies; shard_fake: a gloo double for CPU-only runs).
"""
import ctypes
import os

import numpy as np

import gvom as _gvom

XBUF_SEND_IDS, XBUF_SEND_QUADS, XBUF_SEND_EPS, XBUF_RECV_IDS, XBUF_RECV_QUADS, XBUF_RECV_EPS, XBUF_SEND_RETURNS, XBUF_RECV_RETURNS = range(8)
_I64P = ctypes.POINTER(ctypes.c_int64)


def _nothing():
    return None


def _vec(values):
    a = (ctypes.c_int64 * len(values))(*[int(v) for v in values])
    return a


class HipShardBackend(object):
    """Per-rank compute on one MI355X through the C ABI (include/gvom_hip.h, gvom_shard_* entry points)."""

    def __init__(self, params, rank, world, device, voxel_statistics=False):
        self.rank, self.world = rank, world
        self.g = _gvom.Gvom(*params, device=device, voxel_statistics=voxel_statistics, _shard=(rank, world))
        self.lib, self.h = self.g._lib, self.g._h
        self.xy = params[2]
        self.has_stats = bool(voxel_statistics)
        self.dtype_code = 0                       # cloud type of the scan in flight (0 float32, 1 float64)

    def scan_local(self, pointcloud, ego, tf):
        """This rank's share of the scan -> (send_quads[world], send_eps[world], any_in_grid, n).
        pointcloud: numpy (n, >=3), or (device pointer, n, numpy dtype) for a share already in HBM."""
        g = self.g
        g.ego_position = ego
        on_device = isinstance(pointcloud, tuple)
        if on_device:
            dptr, n, dt = pointcloud
            code = 0 if np.dtype(dt) == np.float32 else 1
            stride = 12 if code == 0 else 24
        else:
            pc, n, stride, code = g._prepare_cloud(pointcloud)
        ego_c = (ctypes.c_double * 3)(float(ego[0]), float(ego[1]), float(ego[2]))
        t = None
        if tf is not None:
            t = np.ascontiguousarray(np.asarray(tf, dtype=np.float64))
            if t.shape != (4, 4):
                raise ValueError("transform must be 4x4")
        sq = (ctypes.c_int64 * self.world)()
        se = (ctypes.c_int64 * self.world)()
        any_ = ctypes.c_int(0)
        src = ctypes.c_void_p(int(dptr)) if on_device else (_gvom._ptr(pc) if n else None)
        g._check(self.lib.gvom_shard_scan_local(self.h, src, 1 if on_device else 0, int(n), stride, code, ego_c,
                                                _gvom._ptr(t), sq, se, ctypes.byref(any_)))
        self.dtype_code = code
        return list(sq), list(se), int(any_.value), n

    def stats_counts(self):
        """returns this rank holds for every other rank's statistics (voxel_statistics handles)"""
        sp = (ctypes.c_int64 * self.world)()
        self.g._check(self.lib.gvom_shard_stats_counts(self.h, sp))
        return list(sp)

    def stats_reserve(self, recv_returns, dtype_code):
        self.g._check(self.lib.gvom_shard_stats_reserve(self.h, _vec(recv_returns), int(dtype_code)))

    def recv_reserve(self, recv_eps):
        self.g._check(self.lib.gvom_shard_recv_reserve(self.h, _vec(recv_eps)))

    def buffer(self, which, peer):
        p, cap = ctypes.c_void_p(), ctypes.c_int64()
        self.g._check(self.lib.gvom_shard_buffer(self.h, which, peer, ctypes.byref(p), ctypes.byref(cap)))
        return p.value, cap.value

    def scan_merge(self, recv_quads, recv_eps, accept):
        self.g._check(self.lib.gvom_shard_scan_merge(self.h, _vec(recv_quads), _vec(recv_eps), 1 if accept else 0))

    def combine_fuse(self):
        return self.g._check(self.lib.gvom_combine_fuse(self.h, None))

    def height_rows(self):
        p, nbytes, rs = ctypes.c_void_p(), ctypes.c_int64(), ctypes.c_int64()
        self.g._check(self.lib.gvom_device_buffer(self.h, _gvom.BUF_HEIGHT_MAPS, ctypes.byref(p), ctypes.byref(nbytes),
                                                  ctypes.byref(rs)))
        return p.value, nbytes.value

    def combine_map2d(self):
        rc, out = self.g._combine_into(self.lib.gvom_combine_map2d_into)
        return out

    def local_fused_cells(self):
        return int(self.g._state().combined_cell_count)

    def set_cell_count(self, n):
        self.g._check(self.lib.gvom_set_combined_cell_count(self.h, int(n)))

    def sync(self):
        self.g._check(self.lib.gvom_sync(self.h))


TRANSPORTS = {"rccl": 0, "peer": 1, "auto": 2, "loopback": 3}
_TRANSPORT_NAMES = {0: "rccl", 1: "peer", 3: "loopback"}


class RcclComm(object):
    """The product transport between rank PROCESSES (libgvom_hip.so: gvom_comm_*): a shared-memory segment for the
    small host-side vectors, and for device data
      transport="rccl"  RCCL over xGMI (grouped ncclSend / ncclRecv, in-place ncclAllGather on the handle's stream),
      transport="peer"  peer copies: exported send regions (hipIpcGetMemHandle) pulled by the receiver with
                        hipMemcpyAsync on its handle's stream -- xGMI between GPUs, and the only transport that runs
                        several ranks on ONE GPU (RCCL refuses that),
      transport="auto"  RCCL, and peer copies on every rank if RCCL cannot initialise on some rank,
      transport="loopback"  RCCL on a ONE-GPU box: the ranks are threads of one process, every rank has a 1-rank
                        communicator and moves what its peers hold for it with ncclSend / ncclRecv to itself on its
                        handle's stream, then the in-place ncclAllGather (include/gvom_hip.h GVOM_TRANSPORT_LOOPBACK).
    One per rank; `name` must be the same on all ranks and unique to this job on the node.  `.transport` says which
    one is in use."""

    def __init__(self, rank, world, device, name, transport="rccl"):
        self.rank, self.world = rank, world
        self.lib = _gvom.load_library()
        self.c = ctypes.c_void_p()
        rc = self.lib.gvom_comm_create2(rank, world, device, name.encode(), TRANSPORTS[transport], ctypes.byref(self.c))
        if rc != 0:
            self.c = ctypes.c_void_p()
            raise _gvom.GvomBackendError("gvom_comm_create2 failed with code %d (rank %d of %d, transport %s)"
                                         % (rc, rank, world, transport))
        self.transport = _TRANSPORT_NAMES[self.lib.gvom_comm_transport(self.c)]
        self.peer_async = bool(self.lib.gvom_comm_peer_async(self.c))       # peer copies without host waits inside an exchange

    def _check(self, rc):
        if rc != 0:
            raise _gvom.GvomBackendError("communicator call failed (%d): %s"
                                         % (rc, self.lib.gvom_comm_last_error(self.c).decode()))

    def peer_stats(self):
        """peer transport: bytes pulled, copies, exports made, refused opens that were repeated"""
        out = (ctypes.c_int64 * 4)()
        self._check(self.lib.gvom_comm_peer_stats(self.c, out))
        d = dict(zip(("bytes", "copies", "exports", "open_retries"), (int(v) for v in out)))
        d["asynchronous"] = self.peer_async
        d["renewed_regions"] = int(self.lib.gvom_comm_peer_renewed(self.c))   # refused exports / opens absorbed by a fresh allocation
        return d

    def info(self):
        """what the communicator itself says: RCCL's own rank count and rank number (None without RCCL), device, PCI bus id"""
        out = (ctypes.c_int64 * 4)()
        bus = ctypes.create_string_buffer(64)
        self._check(self.lib.gvom_comm_info(self.c, out, bus, 64))
        return {"rccl_comm_count": int(out[0]) if out[0] >= 0 else None, "rccl_user_rank": int(out[1]) if out[1] >= 0 else None,
                "device": int(out[2]), "transport": _TRANSPORT_NAMES.get(int(out[3])), "pci_bus_id": bus.value.decode()}

    def wire_stats(self):
        """RCCL calls this rank has issued: ncclSend + ncclRecv calls, their bytes, groups, ncclAllGather calls"""
        out = (ctypes.c_int64 * 4)()
        self._check(self.lib.gvom_comm_wire_stats(self.c, out))
        return dict(zip(("p2p_calls", "p2p_bytes", "groups", "allgathers"), (int(v) for v in out)))

    def abort(self):
        """this rank cannot go on: the other ranks' next wait fails at once instead of timing out"""
        if self.c:
            self.lib.gvom_comm_abort(self.c)

    def exchange_host(self, values):
        k = len(values)
        out = (ctypes.c_int64 * (k * self.world))()
        self._check(self.lib.gvom_comm_exchange_host(self.c, _vec(values), k, out))
        return [list(out[r * k:(r + 1) * k]) for r in range(self.world)]

    # a whole scan / combine in ONE library call (ShardedGvom takes these when the communicator has them: the call-by-call
    # orchestration from Python cost ~25 us per step, profiles/r4_bench_sharded_w1_m256.json against r5's)
    def scan_native(self, backend, pointcloud, ego, tf):
        g = backend.g
        on_device = isinstance(pointcloud, tuple)
        if on_device:
            dptr, n, dt = pointcloud
            code = 0 if np.dtype(dt) == np.float32 else 1
            stride, src = (12 if code == 0 else 24), ctypes.c_void_p(int(dptr))
        else:
            pc, n, stride, code = g._prepare_cloud(pointcloud)
            src = _gvom._ptr(pc) if n else None
        t = None
        if tf is not None:
            t = np.ascontiguousarray(np.asarray(tf, dtype=np.float64))
            if t.shape != (4, 4):
                raise ValueError("transform must be 4x4")
        ego_c = (ctypes.c_double * 3)(float(ego[0]), float(ego[1]), float(ego[2]))
        out = (ctypes.c_int64 * 4)()
        g.ego_position = ego
        self._check(self.lib.gvom_comm_process_pointcloud(self.c, backend.h, src, 1 if on_device else 0, int(n), stride, code, ego_c,
                                                          _gvom._ptr(t), out))
        backend.dtype_code = code
        return bool(out[0]), int(out[1]), (int(out[2]), int(out[3]))

    def combine_native(self, backend):
        lib, c = self.lib, self.c
        rc, out = backend.g._combine_into(lambda h, origin, ptr: self._combine_rc(lib.gvom_comm_combine_maps_into(c, h, origin, ptr)))
        return rc, out

    def _combine_rc(self, rc):
        if rc not in (0, _gvom.GVOM_EMPTY_BUFFER):
            self._check(rc)
        return rc

    def barrier(self):
        self._check(self.lib.gvom_comm_barrier(self.c))

    def before_scan(self):
        """peer transport, asynchronous form: every peer has pulled what the next scan's pack overwrites (else a no-op)"""
        self._check(self.lib.gvom_comm_before_scan(self.c))

    def before_combine(self):
        self._check(self.lib.gvom_comm_before_combine(self.c))

    def exchange_scan(self, backend, send_q, send_e, recv_q, recv_e):
        self._check(self.lib.gvom_comm_exchange_scan(self.c, backend.h, _vec(send_q), _vec(send_e), _vec(recv_q),
                                                     _vec(recv_e)))

    def allgather_rows(self, backend):
        self._check(self.lib.gvom_comm_allgather_rows(self.c, backend.h))

    def exchange_stats(self, backend, send_r, recv_r, bytes_per_return):
        self._check(self.lib.gvom_comm_exchange_stats(self.c, backend.h, _vec(send_r), _vec(recv_r), int(bytes_per_return)))

    def close(self):
        c, self.c = self.c, ctypes.c_void_p()
        if c:
            self.lib.gvom_comm_destroy(c)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class ShardedGvom(object):
    """Same surface as gvom.Gvom (14 positional ctor args, process_pointcloud, combine_maps), for
    `world` cooperating ranks.  Every rank calls every method (SPMD).  process_pointcloud takes THIS
    RANK'S share of the scan (any length, possibly empty); the union of the shares is one logical
    scan, and the result equals gvom.Gvom fed with the concatenated cloud, bit for bit.

    keyword arguments: comm (RcclComm, or a test double with its methods), device, backend (test double),
    voxel_statistics (the reference's per-voxel mean / covariance path, as gvom.Gvom's: every rank also gets the returns
    whose neighbourhood reaches into its rows; make_debug_voxel_map() then returns THIS RANK'S voxels -- the ranks'
    rows together are the unsharded mapper's)."""

    def __init__(self, *params, **kw):
        self.comm = kw.pop("comm")
        self.rank, self.world = self.comm.rank, self.comm.world
        self.params = params
        self.xy_size, self.z_size, self.buffer_size = params[2], params[3], params[4]
        if self.xy_size % (4 * self.world):
            raise ValueError("xy_size (%d) must be a multiple of 4 x the number of ranks (%d)"
                             % (self.xy_size, self.world))
        backend = kw.pop("backend", None)
        device = kw.pop("device", None)
        stats = bool(kw.pop("voxel_statistics", False))
        if backend is None:
            backend = HipShardBackend(params, self.rank, self.world, 0 if device is None else device, stats)
        self.b = backend
        self.ego_position = [0, 0, 0]
        self._cells_dirty = False
        self._cell_count = None
        self.last_exchange_bytes = (0, 0)        # (sent, received) by this rank in the last scan's exchange

    @property
    def combined_cell_count_cpu(self):
        """Global occupied-voxel count of the fused map (gvom.py:217).  Collective: every rank must
        read it (it is not needed on the hot path, so it costs nothing unless asked for)."""
        if self._cells_dirty:
            rows = self.comm.exchange_host([self.b.local_fused_cells()])
            self._cell_count = int(sum(r[0] for r in rows))
            self.b.set_cell_count(self._cell_count)
            self._cells_dirty = False
        return self._cell_count

    def process_pointcloud(self, pointcloud, ego_position, transform=None):
        """pointcloud: this rank's share, numpy (n, >=3) float32 / float64 (or a (device pointer, n, dtype)
        tuple for a share already in HBM); n may differ between ranks."""
        self.ego_position = ego_position
        W, me = self.world, self.rank
        if hasattr(self.comm, "scan_native") and not getattr(self.b, "has_stats", False) and isinstance(self.b, HipShardBackend):
            accept, total_n, self.last_exchange_bytes = self.comm.scan_native(self.b, pointcloud, ego_position, transform)
            if me == 0:
                if total_n == 0:
                    print("[WARNING] Processing an empty pointcloud, nothing will happen!")
                elif not accept:
                    print("[WARNING] The pointcloud points don't overlap with any voxels, nothing will happen!")
            return None
        getattr(self.comm, "before_scan", _nothing)()                # (a transport whose peers may still be reading this rank's send regions)
        send_q, send_e, any_, n = self.b.scan_local(pointcloud, ego_position, transform)
        # one host-side exchange: what every rank packed for every other rank, who saw a return in
        # the grid, how many returns the scan has
        stats = getattr(self.b, "has_stats", False) and W > 1
        send_r = self.b.stats_counts() if stats else []
        table = self.comm.exchange_host(list(send_q) + list(send_e) + [any_, n] + ([self.b.dtype_code] + list(send_r) if stats else []))
        recv_q = [table[s][me] if s != me else 0 for s in range(W)]
        recv_e = [table[s][W + me] if s != me else 0 for s in range(W)]
        accept = any(row[2 * W] for row in table)
        total_n = sum(row[2 * W + 1] for row in table)
        self.b.recv_reserve(recv_e)
        sq = [send_q[d] if d != me else 0 for d in range(W)]
        se = [send_e[d] if d != me else 0 for d in range(W)]
        self.last_exchange_bytes = (1028 * sum(sq) + 8 * sum(se), 1028 * sum(recv_q) + 8 * sum(recv_e))
        self.comm.exchange_scan(self.b, sq, se, recv_q, recv_e)      # the scan's only device exchange
        if stats:
            # the returns whose neighbourhood reaches into another rank's rows (statistics only: the maps do not need them)
            codes = set(row[2 * W + 2] for row in table if row[2 * W + 1] > 0)
            if len(codes) > 1:
                raise ValueError("the ranks' clouds differ in type (float32 / float64)")
            code = codes.pop() if codes else 0
            recv_r = [table[s][2 * W + 3 + me] if s != me else 0 for s in range(W)]
            self.b.stats_reserve(recv_r, code)
            self.comm.exchange_stats(self.b, [send_r[d] if d != me else 0 for d in range(W)], recv_r, 12 if code == 0 else 24)
        self.b.scan_merge(recv_q, recv_e, accept)
        if me == 0:
            if total_n == 0:
                print("[WARNING] Processing an empty pointcloud, nothing will happen!")
            elif not accept:
                print("[WARNING] The pointcloud points don't overlap with any voxels, nothing will happen!")
        return None

    def make_debug_voxel_map(self):
        """this rank's rows of the debug voxel cloud (gvom.py:363-378); None without voxel_statistics"""
        return self.b.g.make_debug_voxel_map()

    def combine_maps(self):
        if hasattr(self.comm, "combine_native") and isinstance(self.b, HipShardBackend):
            rc, out = self.comm.combine_native(self.b)
            if rc == _gvom.GVOM_EMPTY_BUFFER:
                if self.rank == 0:
                    print("[WARNING] The map buffer is empty, nothing will happen!")
                return None
            self._cells_dirty = True
            return out
        getattr(self.comm, "before_combine", _nothing)()             # (... or this rank's rows of the previous combine)
        rc = self.b.combine_fuse()
        if rc == _gvom.GVOM_EMPTY_BUFFER:
            if self.rank == 0:
                print("[WARNING] The map buffer is empty, nothing will happen!")
            return None
        self._cells_dirty = True
        self.comm.allgather_rows(self.b)                              # the combine's only exchange
        return self.b.combine_map2d()


def rendezvous_name():
    """Name of the shared-memory rendezvous for the job this process belongs to.  MASTER_PORT (torchrun and
    bench.py set it) tells jobs on one node apart; the nonce tells this job from an EARLIER one on the same
    port: GVOM_JOB_NONCE if the launcher sets it (bench.py does), else the parent process -- the ranks of one
    job on one node are children of one launcher (torchrun's agent, mpirun's orted, slurmstepd), whose pid
    changes from job to job.  (The library additionally refuses a segment whose creator is no longer alive.)"""
    nonce = os.environ.get("GVOM_JOB_NONCE")
    if nonce is None:
        # the parent names the job only where the ranks are known to be children of ONE launcher process; ranks started from
        # separate shells, per-rank wrapper scripts or one slurmstepd per task have different parents and would wait for each
        # other under different names until the rendezvous times out: they get the fixed name (set GVOM_JOB_NONCE to tell
        # two such jobs on one port apart)
        known = ("TORCHELASTIC_RUN_ID", "OMPI_COMM_WORLD_SIZE", "PMI_SIZE", "PMIX_RANK")
        nonce = "p%d" % os.getppid() if any(k in os.environ for k in known) else "0"
    return "gvom_%s_%s" % (os.environ.get("MASTER_PORT", "29500"), nonce)

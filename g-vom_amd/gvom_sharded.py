"""gvom_sharded -- one G-VOM map sharded over the GPUs of one node (one process per GPU).

Partitioning (DESIGN.md "Multi-GPU"): the voxel grid is stored world-anchored (toroidal),
so its storage y axis is cut into `world` contiguous slabs that NEVER migrate when the
robot-centred window moves.  Rank r owns storage rows [r*xy/world, (r+1)*xy/world) of every
per-voxel array (accumulators, ring slots, fused map).

Per scan (process_pointcloud) -- ONE collective:
    all_gather      every rank's share of the cloud (12 B/point; 1.5 MB per 131k points)
    local           each rank traces the rays that can reach its slab (window-y culling + early
                    exit in k_trace) and commits only voxels of its slab; integer atomics commute,
                    so the result is bit-identical to one GPU.  Every rank sees every point, so
                    each counts the in-grid returns of ALL slabs itself: the reference's "no
                    overlap" test (gvom.py:147-150) needs no collective.
Per combine (combine_maps) -- ONE collective:
    local           temporal fusion + column reductions of the slab -> its rows of the height and
                    inferred-height maps, plus the positive-obstacle density of its cells (the only
                    2-D quantity that needs voxel data)
    all_gather      IN PLACE on the library's row-interleaved buffer [sy][height | inferred |
                    density] (24 B x xy^2 = 1.5 MB at xy=256): the slope stencil needs +-1 row,
                    __guess_height +-15 rows
    local           every rank computes ALL rows of slope / roughness / guess / positive /
                    negative / visibility (20 us of redundant 2-D work instead of a second
                    collective) straight into pinned host memory; every rank returns the maps
There is NO collective on per-voxel data: the only exchanged bytes are the cloud and 2-D rows.
The library runs on the caller's (torch) stream and its own buffer is the collective buffer,
so a combine has no host synchronisation before the final one.

The collectives are torch.distributed calls (backend "nccl" == RCCL over xGMI on ROCm; "gloo"
in the CPU tests); the compute is behind a small backend interface: `HipShardBackend` (the
product: libgvom_hip.so, slab-sharded handle) or a test double injected by tests/.
"""
import ctypes

import numpy as np

import gvom as _gvom


class _DevBuf(object):
    """Exposes a raw device pointer to torch (zero-copy) through __cuda_array_interface__."""

    def __init__(self, ptr, shape, typestr):
        self.__cuda_array_interface__ = {"data": (int(ptr), False), "shape": tuple(shape),
                                         "typestr": typestr, "version": 2}


class HipShardBackend(object):
    """Per-rank compute on one MI355X through the C ABI (include/gvom_hip.h, split entry points).
    The library is attached to torch's current stream and set non-blocking; its own device
    buffer is wrapped as a torch tensor and used directly by the collective."""

    def __init__(self, params, rank, world, device):
        import torch
        self.torch = torch
        self.rank, self.world = rank, world
        self.device = torch.device("cuda", device)
        torch.cuda.set_device(self.device)
        self.g = _gvom.Gvom(*params, device=device, _shard=(rank, world))
        self.lib, self.h = self.g._lib, self.g._h
        self.xy = params[2]
        stream = torch.cuda.current_stream(self.device).cuda_stream
        self.g._check(self.lib.gvom_attach_stream(self.h, ctypes.c_void_p(stream)))
        self.g._check(self.lib.gvom_set_blocking(self.h, 0))
        self.height_full = self._wrap(_gvom.BUF_HEIGHT_MAPS, np.float64)      # [xy, 3*xy]
        self.fused_cells = self._wrap(_gvom.BUF_FUSED_CELLS, np.int64)        # [1]

    def _wrap(self, which, dtype):
        p, nbytes, rs = ctypes.c_void_p(), ctypes.c_int64(), ctypes.c_int64()
        self.g._check(self.lib.gvom_device_buffer(self.h, which, ctypes.byref(p), ctypes.byref(nbytes),
                                                  ctypes.byref(rs)))
        item = np.dtype(dtype).itemsize
        shape = (nbytes.value // rs.value, rs.value // item) if nbytes.value > rs.value else (nbytes.value // item,)
        return self.torch.as_tensor(_DevBuf(p.value, shape, np.dtype(dtype).str), device=self.device)

    def cloud_tensor(self, pc):
        return self.torch.from_numpy(np.ascontiguousarray(pc[:, :3])).to(self.device)

    def process(self, cloud, ego, tf):
        """The whole cloud (torch CUDA tensor (n,3), contiguous, on the attached stream) against
        this rank's slab.  Returns the reference's outcome code (same on every rank)."""
        self.g.ego_position = ego
        return self.g.process_pointcloud_device(cloud.data_ptr(), int(cloud.shape[0]),
                                                np.float32 if cloud.dtype == self.torch.float32 else np.float64,
                                                ego, tf)

    def combine_fuse(self):
        return self.g._check(self.lib.gvom_combine_fuse(self.h, None))

    def set_cell_count(self, n):
        self.g._check(self.lib.gvom_set_combined_cell_count(self.h, int(n)))

    def combine_map2d(self):
        rc, out = self.g._combine_into(self.lib.gvom_combine_map2d_into)
        return out


class ShardedGvom(object):
    """Same surface as gvom.Gvom (14 positional ctor args, process_pointcloud, combine_maps),
    for `world` cooperating ranks.  Every rank calls every method (SPMD).  process_pointcloud
    takes THIS RANK'S share of the scan; the union of the shares is one logical scan, and the
    result equals gvom.Gvom fed with the concatenated cloud, bit for bit."""

    def __init__(self, *params, **kw):
        import torch
        import torch.distributed as dist
        self.torch, self.dist = torch, dist
        self.group = kw.pop("group", None)
        self.rank = dist.get_rank(self.group)
        self.world = dist.get_world_size(self.group)
        self.params = params
        self.xy_size, self.z_size, self.buffer_size = params[2], params[3], params[4]
        if self.xy_size % self.world:
            raise ValueError("xy_size (%d) must be divisible by the number of ranks (%d)"
                             % (self.xy_size, self.world))
        backend = kw.pop("backend", None)
        device = kw.pop("device", None)
        if backend is None:
            backend = HipShardBackend(params, self.rank, self.world, 0 if device is None else device)
        self.b = backend
        self.rows = self.xy_size // self.world
        self.ego_position = [0, 0, 0]
        self._cells_dirty = False
        self._cell_count = None
        self._staged = self.dist.get_backend(self.group) == "gloo"
        self._gather_buf, self._gather_key = None, None

    # -- collectives ------------------------------------------------------------------------
    def _all_gather_cloud(self, local):
        torch = self.torch
        dev = local.device
        if local.is_cuda and self._staged:
            local = local.cpu()          # gloo has no GPU all_gather: stage through the host (tests only)
        key = (local.shape, local.dtype, local.device)
        out = self._gather_buf if self._gather_key == key else None
        if out is None:                  # reused from scan to scan (the trace of the previous scan has
            out = torch.empty((self.world * local.shape[0],) + tuple(local.shape[1:]), dtype=local.dtype,
                              device=local.device)       # finished: the caller saw its return code)
            self._gather_buf, self._gather_key = out, key
        self.dist.all_gather_into_tensor(out, local if local.is_contiguous() else local.contiguous(),
                                         group=self.group)
        return out if out.device == dev else out.to(dev)

    def _all_gather_rows_inplace(self, full):
        """full: [xy, width] tensor whose rows [rank*rows, (rank+1)*rows) are valid on this rank."""
        lo = self.rank * self.rows
        if full.is_cuda and self._staged:
            mine = full[lo:lo + self.rows].cpu()
            out = self.torch.empty((full.shape[0],) + tuple(full.shape[1:]), dtype=full.dtype)
            self.dist.all_gather_into_tensor(out, mine.contiguous(), group=self.group)
            full.copy_(out.to(full.device))
        else:
            self.dist.all_gather_into_tensor(full, full[lo:lo + self.rows], group=self.group)

    @property
    def combined_cell_count_cpu(self):
        """Global occupied-voxel count of the fused map (gvom.py:217).  Collective: every rank
        must read it (it is not needed on the hot path, so it costs nothing unless asked for)."""
        if self._cells_dirty:
            t = self.b.fused_cells.clone()
            if t.is_cuda and self._staged:
                t = t.cpu()
            self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM, group=self.group)
            self._cell_count = int(t.item())
            self.b.set_cell_count(self._cell_count)
            self._cells_dirty = False
        return self._cell_count

    # -- API -----------------------------------------------------------------------------
    def process_pointcloud(self, pointcloud, ego_position, transform=None):
        """pointcloud: this rank's share, numpy (n,>=3) or a torch tensor already on the
        backend's device ((n,3) contiguous; every rank must pass the same n and dtype)."""
        self.ego_position = ego_position
        local = pointcloud if isinstance(pointcloud, self.torch.Tensor) else self.b.cloud_tensor(pointcloud)
        full = self._all_gather_cloud(local)                      # the scan's only collective
        rc = self.b.process(full, ego_position, transform)
        if self.rank == 0:
            if rc == _gvom.GVOM_EMPTY_CLOUD:
                print("[WARNING] Processing an empty pointcloud, nothing will happen!")
            elif rc == _gvom.GVOM_NO_OVERLAP:
                print("[WARNING] The pointcloud points don't overlap with any voxels, nothing will happen!")
        return None

    def combine_maps(self):
        rc = self.b.combine_fuse()
        if rc == _gvom.GVOM_EMPTY_BUFFER:
            if self.rank == 0:
                print("[WARNING] The map buffer is empty, nothing will happen!")
            return None
        self._cells_dirty = True
        self._all_gather_rows_inplace(self.b.height_full)         # the combine's only collective
        return self.b.combine_map2d()

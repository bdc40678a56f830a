"""gvom_sharded -- one G-VOM map sharded over the GPUs of one node (one process per GPU).

Partitioning (DESIGN.md "Multi-GPU"): the voxel grid is stored world-anchored (toroidal),
so its storage y axis is cut into `world` contiguous slabs that NEVER migrate when the
robot-centred window moves.  Rank r owns storage rows [r*xy/world, (r+1)*xy/world) of every
per-voxel array (accumulators, ring slots, fused map) and of every 2-D map.

Per scan (process_pointcloud):
    all_gather      every rank's share of the cloud (N_r x 3 floats; 1.5 MB per 131k points)
    local           each rank traces ALL rays but commits only voxels of its slab
                    (integer atomics commute -> bit-identical to one GPU), encodes its slab
    all_reduce(sum) occupied-voxel count -> the reference's "no overlap" test is global
Per combine (combine_maps):
    local           temporal fusion + column reductions of the slab -> its rows of the
                    height / inferred-height maps
    all_gather      height + inferred rows (2 x 8 B x xy^2: 1 MB at xy=256): the slope stencil
                    needs +-1 row, __guess_height +-15 rows
    local           slope / roughness / guess / positive / negative / visibility rows
    all_gather      the four output maps' rows; every rank returns the full maps
There is NO collective on per-voxel data: the only exchanged bytes are the cloud and 2-D maps.

The collectives are torch.distributed calls (backend "nccl" == RCCL over xGMI on ROCm; "gloo"
in the CPU tests); the compute is behind a small backend interface: `HipShardBackend` (the
product: libgvom_hip.so, slab-sharded handle) or a test double injected by tests/.
"""
import ctypes

import numpy as np

import gvom as _gvom


class HipShardBackend(object):
    """Per-rank compute on one MI355X through the C ABI (include/gvom_hip.h, sharded entry
    points).  Tensors handed to the collectives are torch CUDA tensors; the library copies
    rows device-to-device into/out of them (gvom_rows_export / gvom_rows_import)."""

    def __init__(self, params, rank, world, device):
        import torch
        self.torch = torch
        self.rank, self.world = rank, world
        self.device = torch.device("cuda", device)
        self.g = _gvom.Gvom(*params, device=device, _shard=(rank, world))
        self.lib, self.h = self.g._lib, self.g._h
        self.xy = params[2]
        self.rows = self.xy // world
        self.lo, self.hi = rank * self.rows, (rank + 1) * self.rows

    def empty_rows(self, which, full=False):
        dt = self.torch.float64 if which in (_gvom.MAP_HEIGHT, _gvom.MAP_INFERRED, _gvom.OUT_ROUGHNESS) \
            else self.torch.int32
        return self.torch.empty(((self.xy if full else self.rows), self.xy), dtype=dt, device=self.device)

    def cloud_tensor(self, pc):
        t = self.torch.from_numpy(np.ascontiguousarray(pc[:, :3]))
        return t.to(self.device)

    def scan_begin(self, cloud, ego, tf):
        """cloud: torch CUDA tensor (n,3) f32/f64, contiguous.  Returns (rc, local_cells)."""
        code = 0 if cloud.dtype == self.torch.float32 else 1
        egoc = (ctypes.c_double * 3)(*[float(e) for e in ego])
        tfp = None
        if tf is not None:
            tf = np.ascontiguousarray(np.asarray(tf, np.float64))
            tfp = tf.ctypes.data_as(ctypes.c_void_p)
        cells = ctypes.c_int64(0)
        n = int(cloud.shape[0])
        rc = self.g._check(self.lib.gvom_scan_begin(
            self.h, ctypes.c_void_p(cloud.data_ptr()) if n else None, 1, n, 3 * cloud.element_size(),
            code, egoc, tfp, ctypes.byref(cells)))
        self.g.ego_position = ego
        return rc, int(cells.value)

    def scan_commit(self, accept):
        self.g._check(self.lib.gvom_scan_commit(self.h, 1 if accept else 0))

    def combine_fuse(self):
        cells = ctypes.c_int64(0)
        rc = self.g._check(self.lib.gvom_combine_fuse(self.h, ctypes.byref(cells)))
        return rc, int(cells.value)

    def set_cell_count(self, n):
        self.g._check(self.lib.gvom_set_combined_cell_count(self.h, int(n)))

    def rows_export(self, which):
        t = self.empty_rows(which)
        self.g._check(self.lib.gvom_rows_export(self.h, which, self.lo, self.hi, ctypes.c_void_p(t.data_ptr())))
        return t

    def rows_import(self, which, full):
        self.g._check(self.lib.gvom_rows_import(self.h, which, 0, self.xy, ctypes.c_void_p(full.data_ptr())))

    def combine_map2d(self):
        self.g._check(self.lib.gvom_combine_map2d(self.h))

    def finalize(self):
        xy = self.xy
        origin = np.zeros(3); pos = np.empty((xy, xy), np.int32); neg = np.empty((xy, xy), np.int32)
        rough = np.empty((xy, xy), np.float64); vis = np.empty((xy, xy), np.int32)
        p = _gvom._ptr
        self.g._check(self.lib.gvom_finalize_outputs(self.h, p(origin), p(pos), p(neg), p(rough), p(vis)))
        return origin, pos, neg, rough, vis

    def sync(self):
        self.torch.cuda.synchronize(self.device)


class ShardedGvom(object):
    """Same surface as gvom.Gvom (14 positional ctor args, process_pointcloud, combine_maps),
    for `world` cooperating ranks.  Every rank calls every method (SPMD).  process_pointcloud
    takes THIS RANK'S share of the scan; the union of the shares is one logical scan, and the
    result equals gvom.Gvom fed with the concatenated cloud, bit for bit."""

    def __init__(self, *params, **kw):
        import torch.distributed as dist
        self.dist = dist
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
        self.ego_position = [0, 0, 0]
        self.combined_cell_count_cpu = None

    # -- helpers -------------------------------------------------------------------------
    def _all_gather_rows(self, local):
        torch = __import__("torch")
        dev = local.device
        if local.is_cuda and self.dist.get_backend(self.group) == "gloo":
            local = local.cpu()          # gloo has no GPU all_gather: stage through the host (tests only)
        out = torch.empty((self.world * local.shape[0],) + tuple(local.shape[1:]), dtype=local.dtype,
                          device=local.device)
        self.dist.all_gather_into_tensor(out, local.contiguous(), group=self.group)
        return out.to(dev)

    def _all_reduce_sum(self, value, like):
        torch = __import__("torch")
        dev = like.device if (like is not None and self.dist.get_backend(self.group) != "gloo") else "cpu"
        t = torch.tensor([int(value)], dtype=torch.int64, device=dev)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM, group=self.group)
        return int(t.item())

    # -- API -----------------------------------------------------------------------------
    def process_pointcloud(self, pointcloud, ego_position, transform=None):
        """pointcloud: this rank's share, numpy (n,>=3) or a torch tensor already on the
        backend's device ((n,3) contiguous; every rank must pass the same n and dtype)."""
        torch = __import__("torch")
        self.ego_position = ego_position
        local = pointcloud if isinstance(pointcloud, torch.Tensor) else self.b.cloud_tensor(pointcloud)
        full = self._all_gather_rows(local)                       # exchange step 1: the cloud
        if hasattr(self.b, "sync"):
            self.b.sync()
        rc, cells = self.b.scan_begin(full, ego_position, transform)
        if rc == _gvom.GVOM_EMPTY_CLOUD:
            if self.rank == 0:
                print("[WARNING] Processing an empty pointcloud, nothing will happen!")
            return None
        total = self._all_reduce_sum(cells, full)                 # global "no overlap" test
        self.b.scan_commit(total > 0)
        if total == 0 and self.rank == 0:
            print("[WARNING] The pointcloud points don't overlap with any voxels, nothing will happen!")
        return None

    def combine_maps(self):
        rc, cells = self.b.combine_fuse()
        if rc == _gvom.GVOM_EMPTY_BUFFER:
            if self.rank == 0:
                print("[WARNING] The map buffer is empty, nothing will happen!")
            return None
        h_loc = self.b.rows_export(_gvom.MAP_HEIGHT)
        i_loc = self.b.rows_export(_gvom.MAP_INFERRED)
        total = self._all_reduce_sum(cells, h_loc)
        self.combined_cell_count_cpu = total
        self.b.set_cell_count(total)
        h_full = self._all_gather_rows(h_loc)                     # exchange step 2: height rows
        i_full = self._all_gather_rows(i_loc)
        if hasattr(self.b, "sync"):
            self.b.sync()
        self.b.rows_import(_gvom.MAP_HEIGHT, h_full)
        self.b.rows_import(_gvom.MAP_INFERRED, i_full)
        self.b.combine_map2d()
        for which in (_gvom.OUT_POSITIVE, _gvom.OUT_NEGATIVE, _gvom.OUT_ROUGHNESS, _gvom.OUT_VISIBILITY):
            full = self._all_gather_rows(self.b.rows_export(which))   # exchange step 3: outputs
            if hasattr(self.b, "sync"):
                self.b.sync()
            self.b.rows_import(which, full)
        return self.b.finalize()

"""gvom -- drop-in replacement for the reference module `gvom` on AMD MI355X (gfx950).

    import gvom
    m = gvom.Gvom(xy_resolution, z_resolution, xy_size, z_size, buffer_size, min_distance,
                  positive_obstacle_threshold, negative_obstacle_threshold,
                  slope_obstacle_threshold, robot_height, robot_radius,
                  ground_to_lidar_height, xy_eigen_dist, z_eigen_dist)
    m.process_pointcloud(pc, ego_position, transform)       # gvom_ros.py:109
    origin, positive, negative, roughness, visibility = m.combine_maps()   # gvom_ros.py:115

Same constructor (14 positional args, reference gvom.py:29-31), same methods, return types,
warning strings and ring-buffer attributes as the reference class
(/root/reference/scripts/gvom.py:12-410), so the reference's ROS node (gvom_ros.py) runs
against it unchanged.  All arithmetic happens in hand-written HIP kernels behind the C ABI
of include/gvom_hip.h, bound here with ctypes.  No PyTorch, no Numba.

There is NO CPU fallback: importing works anywhere, but constructing `Gvom` raises
`GvomBackendError` if libgvom_hip.so is missing or no gfx950 device is visible.
"""
import ctypes
import math
import os
import threading

import numpy as np

__all__ = ["Gvom", "GvomBackendError", "load_library", "library_path"]

_HERE = os.path.dirname(os.path.abspath(__file__))

GVOM_OK, GVOM_EMPTY_CLOUD, GVOM_NO_OVERLAP, GVOM_EMPTY_BUFFER, GVOM_NO_DATA = 0, 1, 2, 3, 4
GVOM_ERR_INVALID = -1
GVOM_WHICH_FUSED = -1
MAP_HEIGHT, MAP_INFERRED, MAP_SLOPE_X, MAP_SLOPE_Y, MAP_ROUGHNESS, MAP_GUESSED = range(6)
BUF_HEIGHT_MAPS, BUF_FUSED_CELLS = 0, 3
N_STAGES = 5
STAGE_NAMES = ("trace", "encode", "min_height", "fuse", "map2d")


class GvomBackendError(RuntimeError):
    """The HIP backend is unavailable or a HIP call failed."""


class GvomParams(ctypes.Structure):
    _fields_ = [("xy_resolution", ctypes.c_double), ("z_resolution", ctypes.c_double),
                ("xy_size", ctypes.c_int32), ("z_size", ctypes.c_int32),
                ("buffer_size", ctypes.c_int32), ("reserved0", ctypes.c_int32),
                ("min_distance", ctypes.c_double),
                ("positive_obstacle_threshold", ctypes.c_double),
                ("negative_obstacle_threshold", ctypes.c_double),
                ("slope_obstacle_threshold", ctypes.c_double),
                ("robot_height", ctypes.c_double), ("robot_radius", ctypes.c_double),
                ("ground_to_lidar_height", ctypes.c_double),
                ("xy_eigen_dist", ctypes.c_int32), ("z_eigen_dist", ctypes.c_int32)]


class GvomState(ctypes.Structure):
    _fields_ = [("buffer_index", ctypes.c_int32), ("last_buffer_index", ctypes.c_int32),
                ("has_combined", ctypes.c_int32), ("reserved0", ctypes.c_int32),
                ("combined_cell_count", ctypes.c_int64),
                ("combined_origin", ctypes.c_double * 3), ("ego_position", ctypes.c_double * 3)]


class GvomScanStats(ctypes.Structure):
    _fields_ = [("points", ctypes.c_int64), ("cells", ctypes.c_int64),
                ("sum_hit", ctypes.c_int64), ("sum_total", ctypes.c_int64)]


def library_path():
    return os.environ.get("GVOM_HIP_LIBRARY", os.path.join(_HERE, "lib", "libgvom_hip.so"))


_lib = None
_lib_lock = threading.Lock()

# every symbol include/gvom_hip.h declares: (name, restype, argtypes)
_P, _I, _I64 = ctypes.c_void_p, ctypes.c_int, ctypes.c_int64
_DP = ctypes.POINTER(ctypes.c_double)
ABI = [
    ("gvom_create", _I, [ctypes.POINTER(GvomParams), _I, ctypes.POINTER(_P)]),
    ("gvom_create_sharded", _I, [ctypes.POINTER(GvomParams), _I, _I, _I, ctypes.POINTER(_P)]),
    ("gvom_destroy", None, [_P]),
    ("gvom_process_pointcloud", _I, [_P, _P, _I64, _I64, _I, _DP, _P]),
    ("gvom_process_pointcloud2", _I, [_P, _P, ctypes.c_int64, ctypes.c_int64, ctypes.c_int64, ctypes.c_int64,
                                       ctypes.c_int64, ctypes.c_int, _P, _P]),
    ("gvom_process_pointcloud_device", _I, [_P, _P, _I64, _I64, _I, _DP, _P]),
    ("gvom_combine_maps", _I, [_P, _P, _P, _P, _P, _P]),
    ("gvom_output_buffer_alloc", _I, [_P, ctypes.POINTER(_P)]),
    ("gvom_output_buffer_free", _I, [_P, _P]),
    ("gvom_combine_maps_into", _I, [_P, _P, _P]),
    ("gvom_combine_occupancy_into", _I, [_P, _P, _P, ctypes.c_double, ctypes.c_double, ctypes.c_double]),
    ("gvom_shard_scan_local", _I, [_P, _P, _I, _I64, _I64, _I, _DP, _P, ctypes.POINTER(_I64), ctypes.POINTER(_I64),
                                   ctypes.POINTER(_I)]),
    ("gvom_shard_buffer", _I, [_P, _I, _I, ctypes.POINTER(_P), ctypes.POINTER(_I64)]),
    ("gvom_shard_recv_reserve", _I, [_P, ctypes.POINTER(_I64)]),
    ("gvom_shard_scan_merge", _I, [_P, ctypes.POINTER(_I64), ctypes.POINTER(_I64), _I]),
    ("gvom_shard_stats_counts", _I, [_P, ctypes.POINTER(_I64)]),
    ("gvom_shard_stats_reserve", _I, [_P, ctypes.POINTER(_I64), _I]),
    ("gvom_combine_fuse", _I, [_P, ctypes.POINTER(_I64)]),
    ("gvom_set_combined_cell_count", _I, [_P, _I64]),
    ("gvom_sync", _I, [_P]),
    ("gvom_device_buffer", _I, [_P, _I, ctypes.POINTER(_P), ctypes.POINTER(_I64), ctypes.POINTER(_I64)]),
    ("gvom_combine_map2d_into", _I, [_P, _P, _P]),
    ("gvom_comm_create", _I, [_I, _I, _I, ctypes.c_char_p, ctypes.POINTER(_P)]),
    ("gvom_comm_create2", _I, [_I, _I, _I, ctypes.c_char_p, _I, ctypes.POINTER(_P)]),
    ("gvom_comm_transport", _I, [_P]),
    ("gvom_comm_before_scan", _I, [_P]),
    ("gvom_comm_before_combine", _I, [_P]),
    ("gvom_comm_peer_async", _I, [_P]),
    ("gvom_comm_peer_stats", _I, [_P, ctypes.POINTER(_I64)]),
    ("gvom_comm_peer_renewed", _I64, [_P]),
    ("gvom_comm_info", _I, [_P, ctypes.POINTER(_I64), ctypes.c_char_p, ctypes.c_size_t]),
    ("gvom_comm_wire_stats", _I, [_P, ctypes.POINTER(_I64)]),
    ("gvom_comm_abort", _I, [_P]),
    ("gvom_shard_renew_region", _I, [_P, _I]),
    ("gvom_comm_destroy", None, [_P]),
    ("gvom_comm_exchange_host", _I, [_P, ctypes.POINTER(_I64), _I, ctypes.POINTER(_I64)]),
    ("gvom_comm_barrier", _I, [_P]),
    ("gvom_comm_exchange_scan", _I, [_P, _P, ctypes.POINTER(_I64), ctypes.POINTER(_I64), ctypes.POINTER(_I64),
                                     ctypes.POINTER(_I64)]),
    ("gvom_comm_exchange_stats", _I, [_P, _P, ctypes.POINTER(_I64), ctypes.POINTER(_I64), _I]),
    ("gvom_comm_allgather_rows", _I, [_P, _P]),
    ("gvom_comm_process_pointcloud", _I, [_P, _P, _P, _I, ctypes.c_int64, ctypes.c_int64, _I, _P, _P, _P]),
    ("gvom_comm_combine_maps_into", _I, [_P, _P, _P, _P]),
    ("gvom_comm_rank", _I, [_P]),
    ("gvom_comm_world", _I, [_P]),
    ("gvom_comm_last_error", ctypes.c_char_p, [_P]),
    ("gvom_slot_filled", _I, [_P, _I]),
    ("gvom_get_state", _I, [_P, ctypes.POINTER(GvomState)]),
    ("gvom_get_scan_stats", _I, [_P, ctypes.POINTER(GvomScanStats)]),
    ("gvom_get_occupancy", _I, [_P, _P]),
    ("gvom_debug_voxel_map", _I, [_P, _P, _I64, ctypes.POINTER(_I64)]),
    ("gvom_debug_voxel_eigen", _I, [_P, _P, _P, _I64, ctypes.POINTER(_I64)]),
    ("gvom_read_rows", _I, [_P, _I, _P]),
    ("gvom_gather_metrics", _I, [_P, _I, _P, _I64, _P]),
    ("gvom_debug_height_map", _I, [_P, _P]),
    ("gvom_debug_inferred_height_map", _I, [_P, _P]),
    ("gvom_read_dense", _I, [_P, _I, _P, _P, _P, _P, _P, ctypes.POINTER(_I64)]),
    ("gvom_read_map2d", _I, [_P, _I, _P]),
    ("gvom_last_stage_ms", _I, [_P, ctypes.POINTER(ctypes.c_float * N_STAGES)]),
    ("gvom_combine_begin", _I, [_P, _P, _P]),
    ("gvom_combine_end", _I, [_P, _P]),
    ("gvom_set_profiling", _I, [_P, _I]),
    ("gvom_host_timing", _I, [_P, ctypes.POINTER(ctypes.c_double * 8)]),
    ("gvom_set_tuning", _I, [_P, ctypes.c_char_p, _I]),
    ("gvom_get_tuning", _I, [_P, ctypes.c_char_p, ctypes.POINTER(_I)]),
    ("gvom_stream", _P, [_P]),
    ("gvom_alloc_generation", ctypes.c_uint64, [_P]),
    ("gvom_region_generation", ctypes.c_uint64, [_P, _I]),
    ("gvom_last_error", ctypes.c_char_p, [_P]),
    ("gvom_backend_info", _I, [ctypes.c_char_p, ctypes.c_size_t]),
    ("gvom_abi_version", _I, []),
]


ABI_VERSION = 8          # include/gvom_hip.h GVOM_ABI_VERSION this binding was written against


def load_library(path=None):
    """dlopen libgvom_hip.so and bind every C-ABI entry point.  Raises GvomBackendError."""
    global _lib
    with _lib_lock:
        if _lib is not None and path is None:
            return _lib
        p = path or library_path()
        if not os.path.exists(p):
            raise GvomBackendError(
                "HIP backend not built: %s is missing. Run `make -C %s` (or "
                "`python -c 'import __graft_entry__ as g; g.build()'`). There is no CPU fallback."
                % (p, _HERE))
        try:
            L = ctypes.CDLL(p)
        except OSError as e:
            raise GvomBackendError("cannot load %s: %s" % (p, e))
        for name, res, args in ABI:
            try:
                f = getattr(L, name)
            except AttributeError:
                raise GvomBackendError("%s does not export %s" % (p, name))
            f.restype = res
            f.argtypes = args
        if L.gvom_abi_version() != ABI_VERSION:
            raise GvomBackendError("%s has C ABI version %d, this binding needs %d: rebuild it (`make -C %s`)"
                                   % (p, L.gvom_abi_version(), ABI_VERSION, _HERE))
        if path is None:
            _lib = L
        return L


def _ptr(a):
    return None if a is None else a.ctypes.data_as(ctypes.c_void_p)


class _DeviceArrayView(object):
    """Stands in for the reference's numba device arrays: `.copy_to_host()` returns numpy."""

    def __init__(self, fetch):
        self._fetch = fetch

    def copy_to_host(self):
        return self._fetch()

    def __array__(self, dtype=None):
        a = self._fetch()
        return a if dtype is None else a.astype(dtype)


class _OutputPool(object):
    """Free pinned output buffers of one Gvom.  Outlives the Gvom if returned arrays do: a buffer that
    comes back after the mapper is gone is released at once (pinned host memory is not tied to the
    handle: hipHostFree works without it)."""

    def __init__(self):
        self.free = []
        self.closed = False

    def give_back(self, ptr):
        if not self.closed:
            self.free.append(ptr)           # recycled by the next combine_maps
        else:
            _host_free(ptr)


def _host_free(ptr):
    try:
        rt = ctypes.CDLL("libamdhip64.so")
        rt.hipHostFree.argtypes = [ctypes.c_void_p]
        rt.hipHostFree(ctypes.c_void_p(ptr))
    except Exception:
        pass


class _PinnedOutput(object):
    """One pinned, device-mapped output buffer of combine_maps.  The four returned numpy arrays
    are views whose base chain ends here; when the caller drops them all, the buffer goes back to
    the owning Gvom's pool (so every call still returns FRESH arrays, as the reference does) -- or is
    freed, if that Gvom no longer exists."""

    def __init__(self, owner_pool, ptr, nbytes):
        self._pool = owner_pool
        self.ptr = ptr
        self.__array_interface__ = {"data": (ptr, False), "shape": (nbytes,), "typestr": "|u1",
                                    "version": 3}

    def __del__(self):
        pool = self._pool
        if pool is not None:
            pool.give_back(self.ptr)


def transform_from_translation_rotation(translation, rotation):
    """4x4 matrix of a translation (x, y, z) and a quaternion (x, y, z, w): what the node builds with
    tf.TransformerROS.fromTranslationRotation (gvom_ros.py:105).  Restates tf.transformations
    (ROS geometry 1.13, not part of the reference checkout): translation_matrix @ quaternion_matrix,
    with quaternion_matrix's normalisation q *= sqrt(2 / q.q) and its near-zero quaternion -> identity."""
    q = np.array(rotation[:4], dtype=np.float64, copy=True)
    nq = np.dot(q, q)
    if nq < np.finfo(float).eps * 4.0:
        r = np.identity(4)
    else:
        q *= np.sqrt(2.0 / nq)
        q = np.outer(q, q)
        r = np.array(((1.0 - q[1, 1] - q[2, 2], q[0, 1] - q[2, 3], q[0, 2] + q[1, 3], 0.0),
                      (q[0, 1] + q[2, 3], 1.0 - q[0, 0] - q[2, 2], q[1, 2] - q[0, 3], 0.0),
                      (q[0, 2] - q[1, 3], q[1, 2] + q[0, 3], 1.0 - q[0, 0] - q[1, 1], 0.0),
                      (0.0, 0.0, 0.0, 1.0)), dtype=np.float64)
    t = np.identity(4)
    t[:3, 3] = translation[:3]
    return np.dot(t, r)


class _PendingMaps(object):
    """A combine begun with Gvom.combine_maps_async() / combine_maps_occupancy_async(); result() completes
    it (once)."""

    def __init__(self, owner, holder, occupancy=False):
        self._owner, self._holder, self._out, self._done = owner, holder, None, holder is None
        self._occupancy = occupancy

    def result(self):
        if not self._done:
            g = self._owner
            xy = g.xy_size
            n2 = xy * xy
            origin = np.zeros(3, np.float64)
            g._check(g._lib.gvom_combine_end(g._h, _ptr(origin)))
            raw = np.asarray(self._holder)
            if self._occupancy:
                self._out = (origin,) + tuple(raw[k * n2:(k + 1) * n2].view(np.int8) for k in range(5))
            else:
                self._out = (origin,
                             np.ndarray((xy, xy), np.int32, raw, 0, (4, 4 * xy)),
                             np.ndarray((xy, xy), np.int32, raw, 4 * n2, (4, 4 * xy)),
                             np.ndarray((xy, xy), np.float64, raw, 12 * n2, (8, 8 * xy)),
                             np.ndarray((xy, xy), np.int32, raw, 8 * n2, (4, 4 * xy)))
            self._done, self._holder = True, None
        return self._out

    def __del__(self):
        # a handle dropped without result(): end the combine so that the mapper accepts the next one
        try:
            if not self._done and self._owner._h:
                self._owner._lib.gvom_combine_end(self._owner._h, None)
        except Exception:
            pass


class Gvom(object):
    """A class to convert lidar pointclouds into a cost map (reference gvom.py:12-27).

    The 14 positional arguments are the reference's (gvom.py:29-31).  Keyword arguments of this implementation:
      device            HIP device of the map (default 0)
      voxel_statistics  the per-voxel mean / covariance path behind make_debug_voxel_map (gvom.py:159, 276-284, 363-378).
                        None (default): ON DEMAND -- it runs from the first scan on, as in the reference, for as long as somebody
                        reads it (make_debug_voxel_map, metrics_buffer, combined_metrics, voxels_eigenvalues: the unchanged
                        node does every tick, gvom_ros.py:171); three combines in a row without a read switch it off (the
                        scans then cost what the north-star path costs), a later read returns None once and switches it on
                        again for the scans that follow.  True: always.  False: never (make_debug_voxel_map returns None).
      c_order           False (default): combine_maps returns the four maps as FORTRAN-ordered views of pinned host memory the
                        GPU has written -- same [x, y] indexing, shapes, dtypes and values as the reference's arrays; its
                        caller flattens them with order='F' (gvom_ros.py:141-162), which is then a no-copy reshape.  True:
                        C-contiguous arrays of their own, as the reference's copy_to_host() returns (gvom.py:352-354).
      numba_cuda_typing the types Numba infers for a REAL CUDA device where they differ from its simulator's (which the golden
                        fixtures were recorded under): ray_length = sqrt(float32) in float32, slope / ray_length in float32,
                        the loop bound from that float32 (gvom.py:1109-1114, 1127; profiles/numba_cuda_typing.txt, INTEGRATION.md
                        section 5).  cuda_f32_sqrt: the name this switch had before round 6."""

    def __init__(self, xy_resolution, z_resolution, xy_size, z_size, buffer_size, min_distance,
                 positive_obstacle_threshold, negative_obstacle_threshold, slope_obstacle_threshold,
                 robot_height, robot_radius, ground_to_lidar_height, xy_eigen_dist, z_eigen_dist,
                 device=0, voxel_statistics=None, numba_cuda_typing=False, c_order=False, _shard=None, _library=None, cuda_f32_sqrt=None):
        self.xy_resolution = xy_resolution
        self.z_resolution = z_resolution
        self.xy_size = xy_size
        self.z_size = z_size
        self.voxel_count = self.xy_size * self.xy_size * self.z_size
        self.min_distance = min_distance
        self.positive_obstacle_threshold = positive_obstacle_threshold
        self.negative_obstacle_threshold = negative_obstacle_threshold
        self.slope_obstacle_threshold = slope_obstacle_threshold
        self.robot_height = robot_height
        self.robot_radius = robot_radius
        self.ground_to_lidar_height = ground_to_lidar_height
        self.xy_eigen_dist = xy_eigen_dist
        self.z_eigen_dist = z_eigen_dist
        self.metrics_count = 10
        self.buffer_size = buffer_size
        self.threads_per_block = 256
        self.threads_per_block_3D = (8, 8, 4)
        self.threads_per_block_2D = (16, 16)
        self.blocks = math.ceil(self.voxel_count / self.threads_per_block)          # gvom.py:94
        self.ego_position = [0, 0, 0]
        # reference attributes that guard ITS buffers (gvom.py:65-67, 96) or hold a placeholder (gvom.py:54): inert here -- the
        # library serialises per handle with its own mutex -- but present, for callers that touch them
        self.semaphores = [threading.Semaphore() for _ in range(int(buffer_size))]
        self.ego_semaphore = threading.Semaphore()
        self.metrics = _DeviceArrayView(lambda: np.array([[3, 2]]))
        self._c_order = bool(c_order)
        if cuda_f32_sqrt is not None:
            numba_cuda_typing = bool(cuda_f32_sqrt)

        self._lib = load_library(_library)
        self._h = ctypes.c_void_p()
        stat_flags = 4 if (voxel_statistics is None and _shard is None) else (1 if voxel_statistics else 0)
        prm = GvomParams(float(xy_resolution), float(z_resolution), int(xy_size), int(z_size),
                         int(buffer_size), stat_flags | (2 if numba_cuda_typing else 0), float(min_distance),
                         float(positive_obstacle_threshold), float(negative_obstacle_threshold),
                         float(slope_obstacle_threshold), float(robot_height), float(robot_radius),
                         float(ground_to_lidar_height), int(xy_eigen_dist), int(z_eigen_dist))
        if _shard is None:
            rc = self._lib.gvom_create(ctypes.byref(prm), int(device), ctypes.byref(self._h))
        else:
            rc = self._lib.gvom_create_sharded(ctypes.byref(prm), int(device), int(_shard[0]),
                                               int(_shard[1]), ctypes.byref(self._h))
        self._out_pool = _OutputPool()      # free pinned output buffers (host pointers)
        if rc != GVOM_OK:
            info = ctypes.create_string_buffer(256)
            self._lib.gvom_backend_info(info, 256)
            self._h = ctypes.c_void_p()
            raise GvomBackendError("gvom_create failed with code %d (%s). There is no CPU fallback."
                                   % (rc, info.value.decode()))

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            try:
                # idle buffers go now; those still referenced by live arrays are freed by the pool when
                # the last array of their call is collected
                pool = getattr(self, "_out_pool", None)
                if pool is not None:
                    pool.closed = True
                    for p in pool.free:
                        self._lib.gvom_output_buffer_free(h, ctypes.c_void_p(p))
                    pool.free = []
                self._lib.gvom_destroy(h)
            except Exception:
                pass

    # ------------------------------------------------------------------------------------
    def _check(self, rc):
        if rc < 0:
            raise GvomBackendError("libgvom_hip call failed (%d): %s"
                                   % (rc, self._lib.gvom_last_error(self._h).decode()))
        return rc

    @staticmethod
    def _prepare_cloud(pointcloud):
        """(array, n, row_stride_bytes, dtype_code).  float32/float64 rows of >= 3 columns are
        passed without a copy when C-contiguous in the last axis; anything else is converted to
        float64 (what ros_numpy delivers, gvom_ros.py:108)."""
        pc = pointcloud if isinstance(pointcloud, np.ndarray) else np.asarray(pointcloud)
        if pc.ndim != 2 or pc.shape[1] < 3:
            raise ValueError("pointcloud must have shape (N, >=3), got %r" % (pc.shape,))
        if pc.dtype not in (np.float32, np.float64):
            pc = pc.astype(np.float64)
        if pc.shape[0] > 0 and (pc.strides[1] != pc.itemsize or pc.strides[0] < 3 * pc.itemsize
                                or pc.strides[0] % pc.itemsize):
            pc = np.ascontiguousarray(pc)
        stride = pc.strides[0] if pc.shape[0] > 0 else 3 * pc.itemsize
        return pc, pc.shape[0], stride, (0 if pc.dtype == np.float32 else 1)

    def process_pointcloud(self, pointcloud, ego_position, transform=None):
        """Imports a pointcloud, processes it into a voxel map then adds the map to the buffer
        (reference gvom.py:99-175).  Returns None."""
        self.ego_position = ego_position
        pc, n, stride, code = self._prepare_cloud(pointcloud)
        ego = (ctypes.c_double * 3)(float(ego_position[0]), float(ego_position[1]),
                                    float(ego_position[2]))
        tf = None
        if transform is not None:
            tf = np.ascontiguousarray(np.asarray(transform, dtype=np.float64))
            if tf.shape != (4, 4):
                raise ValueError("transform must be 4x4")
        rc = self._check(self._lib.gvom_process_pointcloud(self._h, _ptr(pc) if n else None, n,
                                                           stride, code, ego, _ptr(tf)))
        if rc == GVOM_EMPTY_CLOUD:
            print("[WARNING] Processing an empty pointcloud, nothing will happen!")
        elif rc == GVOM_NO_OVERLAP:
            print("[WARNING] The pointcloud points don't overlap with any voxels, nothing will happen!")
        return None

    def process_pointcloud_device(self, dev_ptr, n, dtype, ego_position, transform=None,
                                  row_stride_bytes=None):
        """Same as process_pointcloud for a cloud already resident in HBM (raw device pointer)."""
        self.ego_position = ego_position
        code = 0 if (dtype is np.float32 or np.dtype(dtype) == np.float32) else 1
        stride = row_stride_bytes or (12 if code == 0 else 24)
        ego = (ctypes.c_double * 3)(float(ego_position[0]), float(ego_position[1]),
                                    float(ego_position[2]))
        tf = None
        if transform is not None:
            tf = np.ascontiguousarray(np.asarray(transform, dtype=np.float64))
        return self._check(self._lib.gvom_process_pointcloud_device(
            self._h, ctypes.c_void_p(int(dev_ptr)), int(n), int(stride), code, ego, _ptr(tf)))

    # ---- ingest side of the ROS node (reference gvom_ros.py:93-109) ----------
    def process_pointcloud2(self, data, n_points, point_step, offsets, ego_position, transform=None,
                            field_dtype=np.float32):
        """Scans the packed bytes of a sensor_msgs/PointCloud2 directly: `data` (bytes / buffer) holds
        n_points records of point_step bytes with little-endian x, y, z fields of `field_dtype`
        (float32 = PointField.FLOAT32, float64 = FLOAT64) at byte `offsets` (x, y, z).  Equivalent to
            pc = ros_numpy.point_cloud2.pointcloud2_to_xyz_array(msg)     # gvom_ros.py:108
            self.process_pointcloud(pc, ego_position, transform)          # gvom_ros.py:109
        (ros_numpy hands over a float64 array with the non-finite records removed; here FLOAT32
        fields are widened on the GPU and non-finite records have no effect on the map)."""
        buf = np.frombuffer(data, dtype=np.uint8)
        if buf.size < int(n_points) * int(point_step):
            raise ValueError("PointCloud2 data shorter than n_points * point_step")
        code = 0 if np.dtype(field_dtype) == np.float32 else 1
        ego = (ctypes.c_double * 3)(float(ego_position[0]), float(ego_position[1]), float(ego_position[2]))
        self.ego_position = ego_position
        tf = None
        if transform is not None:
            tf = np.ascontiguousarray(np.asarray(transform, dtype=np.float64))
        rc = self._check(self._lib.gvom_process_pointcloud2(
            self._h, ctypes.c_void_p(buf.ctypes.data), int(n_points), int(point_step),
            int(offsets[0]), int(offsets[1]), int(offsets[2]), code, ego, _ptr(tf)))
        if rc == GVOM_EMPTY_CLOUD:
            print("[WARNING] Processing an empty pointcloud, nothing will happen!")
        elif rc == GVOM_NO_OVERLAP:
            print("[WARNING] The pointcloud points don't overlap with any voxels, nothing will happen!")
        return None

    def process_pointcloud2_msg(self, msg, ego_position, transform=None):
        """process_pointcloud2 for a sensor_msgs/PointCloud2-like object (fields[].name/offset/datatype,
        point_step, row_step, width, height, data, is_bigendian)."""
        f = {fd.name: fd for fd in msg.fields}
        kinds = {f[k].datatype for k in "xyz"}
        if getattr(msg, "is_bigendian", False) or len(kinds) != 1 or not kinds <= {7, 8}:
            raise ValueError("x, y, z must be little-endian FLOAT32 (7) or FLOAT64 (8) fields of one type")
        if msg.height > 1 and msg.row_step != msg.width * msg.point_step:
            raise ValueError("row padding is not supported")
        return self.process_pointcloud2(msg.data, msg.width * msg.height, msg.point_step,
                                        (f["x"].offset, f["y"].offset, f["z"].offset), ego_position,
                                        transform, np.float32 if kinds == {7} else np.float64)

    def combine_maps(self):
        """Combines all maps in the buffer and processes the resultant map into 2D maps
        (reference gvom.py:177-354).  Returns None or (origin_world f64[3], positive i32[xy,xy],
        negative i32[xy,xy], roughness f64[xy,xy], visibility i32[xy,xy])."""
        if self._c_order:
            xy = self.xy_size
            origin = np.zeros(3, np.float64)
            positive, negative, visibility = (np.empty((xy, xy), np.int32) for _ in range(3))
            roughness = np.empty((xy, xy), np.float64)
            rc = self._check(self._lib.gvom_combine_maps(self._h, _ptr(origin), _ptr(positive), _ptr(negative), _ptr(roughness),
                                                         _ptr(visibility)))
            out = (origin, positive, negative, roughness, visibility)
        else:
            rc, out = self._combine_into(self._lib.gvom_combine_maps_into)
        if rc == GVOM_EMPTY_BUFFER:
            print("[WARNING] The map buffer is empty, nothing will happen!")
            return None
        return out

    def combine_maps_async(self):
        """combine_maps() split in two (an extension; the reference's call is synchronous): enqueues the
        combine and returns a handle at once; `.result()` waits and returns what combine_maps() returns.
        Hand the next scan to process_pointcloud*() in between: its ray tracing runs on the GPU while the
        maps of this combine are written to host memory.  One combine may be pending at a time."""
        xy = self.xy_size
        n2 = xy * xy
        if self._out_pool.free:
            ptr = self._out_pool.free.pop()
        else:
            p = ctypes.c_void_p()
            self._check(self._lib.gvom_output_buffer_alloc(self._h, ctypes.byref(p)))
            ptr = p.value
        holder = _PinnedOutput(self._out_pool, ptr, n2 * 20)
        rc = self._check(self._lib.gvom_combine_begin(self._h, ctypes.c_void_p(ptr), None))
        if rc == GVOM_EMPTY_BUFFER:
            print("[WARNING] The map buffer is empty, nothing will happen!")
            return _PendingMaps(self, None)
        return _PendingMaps(self, holder)

    def combine_maps_occupancy_async(self, density_threshold=50, min_roughness=-10, max_roughness=0):
        """combine_maps_occupancy() split like combine_maps_async(): `.result()` returns its tuple."""
        n2 = self.xy_size * self.xy_size
        if self._out_pool.free:
            ptr = self._out_pool.free.pop()
        else:
            p = ctypes.c_void_p()
            self._check(self._lib.gvom_output_buffer_alloc(self._h, ctypes.byref(p)))
            ptr = p.value
        holder = _PinnedOutput(self._out_pool, ptr, n2 * 20)
        occ = (ctypes.c_double * 3)(float(density_threshold), float(min_roughness), float(max_roughness))
        rc = self._check(self._lib.gvom_combine_begin(self._h, ctypes.c_void_p(ptr), occ))
        if rc == GVOM_EMPTY_BUFFER:
            print("[WARNING] The map buffer is empty, nothing will happen!")
            return _PendingMaps(self, None)
        return _PendingMaps(self, holder, occupancy=True)

    def combine_maps_occupancy(self, density_threshold=50, min_roughness=-10, max_roughness=0):
        """combine_maps() fused with the post-processing the ROS node applies to its result
        (reference gvom_ros.py:141-165).  Advances the map exactly like
        combine_maps() and returns None (empty ring) or
            (origin_world f64[3], hard, soft, certainty, negative, roughness)
        where each grid is the int8[xy*xy] array the node assigns to nav_msgs/OccupancyGrid.data
        (x fastest, i.e. np.reshape(m, -1, order='F')).  Defaults = the node's ROS parameter
        defaults (gvom_ros.py:32-35).  5 bytes per cell cross PCIe instead of 20."""
        xy = self.xy_size
        n2 = xy * xy
        origin = np.zeros(3, np.float64)
        if self._out_pool.free:
            ptr = self._out_pool.free.pop()
        else:
            p = ctypes.c_void_p()
            self._check(self._lib.gvom_output_buffer_alloc(self._h, ctypes.byref(p)))
            ptr = p.value
        holder = _PinnedOutput(self._out_pool, ptr, n2 * 20)
        rc = self._check(self._lib.gvom_combine_occupancy_into(
            self._h, _ptr(origin), ctypes.c_void_p(ptr), float(density_threshold),
            float(min_roughness), float(max_roughness)))
        if rc == GVOM_EMPTY_BUFFER:
            print("[WARNING] The map buffer is empty, nothing will happen!")
            return None
        raw = np.asarray(holder)
        grids = tuple(raw[k * n2:(k + 1) * n2].view(np.int8) for k in range(5))
        return (origin,) + grids

    def _combine_into(self, entry_point):
        """Runs `entry_point(handle, origin, pinned_buffer)` and wraps the pinned buffer as the
        reference's return tuple.  The GPU writes the four maps straight into a pinned,
        device-mapped host buffer; the returned arrays are views of it (fresh per call: a buffer
        is reused only after every array of an earlier call has been garbage-collected)."""
        xy = self.xy_size
        n2 = xy * xy
        origin = np.zeros(3, np.float64)
        if self._out_pool.free:
            ptr = self._out_pool.free.pop()
        else:
            p = ctypes.c_void_p()
            self._check(self._lib.gvom_output_buffer_alloc(self._h, ctypes.byref(p)))
            ptr = p.value
        holder = _PinnedOutput(self._out_pool, ptr, n2 * 20)
        # The views are built BEFORE the (blocking) call: k_encode is still running on the GPU when
        # combine_maps is entered, so this host work is hidden; after the call only the return is left.
        raw = np.asarray(holder)
        # the library writes the maps in [y][x] memory order: seen through .T they are the reference's
        # [x, y]-indexed arrays in Fortran order (what gvom_ros.py's reshape(..., order='F') reads
        # without a copy), and the GPU writes them as contiguous runs without a transpose
        positive = np.ndarray((xy, xy), np.int32, raw, 0, (4, 4 * xy))
        negative = np.ndarray((xy, xy), np.int32, raw, 4 * n2, (4, 4 * xy))
        visibility = np.ndarray((xy, xy), np.int32, raw, 8 * n2, (4, 4 * xy))
        roughness = np.ndarray((xy, xy), np.float64, raw, 12 * n2, (8, 8 * xy))
        out = (origin, positive, negative, roughness, visibility)
        rc = self._check(entry_point(self._h, _ptr(origin), ctypes.c_void_p(ptr)))
        if rc != GVOM_OK:
            return rc, None
        return GVOM_OK, out

    # ---- accessors / debug API (reference gvom.py:356-410) ------------------------------
    def get_map_as_occupancy_grid(self):
        out = np.empty((self.xy_size, self.xy_size, self.z_size), np.uint8)
        rc = self._check(self._lib.gvom_get_occupancy(self._h, _ptr(out)))
        if rc == GVOM_NO_DATA:
            raise AttributeError("'NoneType' object has no attribute 'copy_to_host'")  # as the reference
        return out.astype(bool)

    def make_debug_voxel_map(self):
        """float32[Cc, 8] rows {x, y, z, hit/total, hit, l0-l1, l1-l2, l2} (reference gvom.py:363-378) while the mapper
        computes the per-voxel statistics (voxel_statistics: by default for as long as this is called); else None, which
        the reference's caller tolerates (gvom_ros.py:171-172) -- and which switches them on again for the scans that follow."""
        n = self.combined_cell_count_cpu
        if n is None:
            print("No data")
            return None
        out = np.empty((max(n, 1), 8), np.float32)
        rows = ctypes.c_int64(0)
        rc = self._check(self._lib.gvom_debug_voxel_map(self._h, _ptr(out), n, ctypes.byref(rows)))
        if rc == GVOM_NO_DATA:
            return None
        return out[:min(n, int(rows.value))]

    def _rows(self, which):
        rows = np.empty(self.voxel_count, np.int32)
        rc = self._check(self._lib.gvom_read_rows(self._h, int(which), _ptr(rows)))
        return None if rc == GVOM_NO_DATA else rows[rows >= 0]          # occupied voxels, in voxel order

    def _metrics(self, which, dtype):
        """(C, 10) statistics {mean xyz, covariance xx xy xz yy yz zz, count} of the occupied voxels of a
        ring slot (float64) or of the fused map (float32), rows in voxel order (row order is unspecified
        in the reference, gvom.py:1158); None without voxel statistics."""
        rows = self._rows(which)
        if rows is None:
            return None
        out = np.empty((rows.shape[0], 10), dtype)
        rc = self._check(self._lib.gvom_gather_metrics(self._h, int(which), _ptr(np.ascontiguousarray(rows)),
                                                       rows.shape[0], _ptr(out)))
        return None if rc == GVOM_NO_DATA else out

    @property
    def metrics_buffer(self):
        """reference attribute (gvom.py:62,166): per ring slot None or a device-array stand-in of (C, 10) float64"""
        out = []
        for i in range(self.buffer_size):
            if self._lib.gvom_slot_filled(self._h, i) != 1 or self._metrics(i, np.float64) is None:
                out.append(None)
            else:
                out.append(_DeviceArrayView(lambda i=i: self._metrics(i, np.float64)))
        return out

    @property
    def combined_metrics(self):
        """reference attribute (gvom.py:72,234): (Cc, 10) float32 of the fused map, rows in voxel order"""
        if not self._state().has_combined or self._metrics(GVOM_WHICH_FUSED, np.float32) is None:
            return None
        return _DeviceArrayView(lambda: self._metrics(GVOM_WHICH_FUSED, np.float32))

    last_combined_metrics = combined_metrics

    @property
    def voxels_eigenvalues(self):
        """reference attribute (gvom.py:83,281, set by make_debug_voxel_map): (Cc, 3) float32 eigenvalues
        l0 >= l1 >= l2 of the fused voxels' covariances, rows in voxel order"""
        n = self.combined_cell_count_cpu
        if n is None:
            return None
        out = np.empty((max(n, 1), 8), np.float32)
        eig = np.empty((max(n, 1), 3), np.float32)
        rows = ctypes.c_int64(0)
        rc = self._check(self._lib.gvom_debug_voxel_eigen(self._h, _ptr(out), _ptr(eig), n, ctypes.byref(rows)))
        if rc == GVOM_NO_DATA:
            return None
        k = min(n, int(rows.value))
        st = self._state()
        # voxel order: the rows carry their world coordinates (gvom.py:462-466)
        x = np.rint(out[:k, 0] / self.xy_resolution - st.combined_origin[0]).astype(np.int64)
        y = np.rint(out[:k, 1] / self.xy_resolution - st.combined_origin[1]).astype(np.int64)
        z = np.rint(out[:k, 2] / self.z_resolution - st.combined_origin[2]).astype(np.int64)
        order = np.argsort(x + y * self.xy_size + z * self.xy_size * self.xy_size, kind="stable")
        arr = np.ascontiguousarray(eig[:k][order])
        return _DeviceArrayView(lambda: arr.copy())

    def make_debug_height_map(self):
        out = np.empty((self.xy_size * self.xy_size, 7), np.float32)
        rc = self._check(self._lib.gvom_debug_height_map(self._h, _ptr(out)))
        if rc == GVOM_NO_DATA:
            print("No data")
            return None
        return out

    def make_debug_inferred_height_map(self):
        out = np.empty((self.xy_size * self.xy_size, 3), np.float32)
        rc = self._check(self._lib.gvom_debug_inferred_height_map(self._h, _ptr(out)))
        if rc == GVOM_NO_DATA:
            print("No data")
            return None
        return out

    # ---- ring-buffer / fused-map attributes of the reference object ------------------
    def _state(self):
        st = GvomState()
        self._check(self._lib.gvom_get_state(self._h, ctypes.byref(st)))
        return st

    @property
    def buffer_index(self):
        return int(self._state().buffer_index)

    @property
    def last_buffer_index(self):
        return int(self._state().last_buffer_index)

    @property
    def combined_cell_count_cpu(self):
        st = self._state()
        return int(st.combined_cell_count) if st.has_combined else None

    last_combined_cell_count_cpu = combined_cell_count_cpu

    def read_dense(self, which):
        """Test hook: (state, hit, total, min_h, origin, cell_count) dense arrays in the
        reference's voxel order for ring slot `which` or GVOM_WHICH_FUSED; None if empty."""
        V = self.voxel_count
        state = np.empty(V, np.int32); hit = np.empty(V, np.int32); total = np.empty(V, np.int32)
        minh = np.empty(V, np.float32); origin = np.zeros(3); cnt = ctypes.c_int64(0)
        rc = self._check(self._lib.gvom_read_dense(self._h, int(which), _ptr(state), _ptr(hit),
                                                   _ptr(total), _ptr(minh), _ptr(origin),
                                                   ctypes.byref(cnt)))
        if rc == GVOM_NO_DATA:
            return None
        return state, hit, total, minh, origin, int(cnt.value)

    def _compact(self, which):
        """Reference-shaped sparse form (index_map, hit[C], total[C], min_height) rebuilt from
        the dense test hook; rows numbered in voxel order (row order is unspecified in the
        reference, gvom.py:1158)."""
        d = self.read_dense(which)
        if d is None:
            return None
        state, hit, total, minh, origin, _ = d
        occ = state >= 0
        index_map = state.copy()
        index_map[occ] = np.arange(int(occ.sum()), dtype=np.int32)
        return index_map, hit[occ], total[occ], minh[occ], origin

    def _slot_views(self, field):
        out = []
        for i in range(self.buffer_size):
            if self._lib.gvom_slot_filled(self._h, i) != 1:
                out.append(None)
                continue

            def fetch(i=i, field=field):
                index_map, hit, total, minh, origin = self._compact(i)
                if field == 4:                       # gvom.py:1014: min_height has 3*C entries
                    return np.concatenate([minh, np.ones(2 * minh.shape[0], np.float32)])
                return (index_map, hit, total, minh, origin)[field]
            out.append(_DeviceArrayView(fetch))
        return out

    index_buffer = property(lambda self: self._slot_views(0))
    hit_count_buffer = property(lambda self: self._slot_views(1))
    total_count_buffer = property(lambda self: self._slot_views(2))
    min_height_buffer = property(lambda self: self._slot_views(4))

    @property
    def origin_buffer(self):
        out = []
        for i in range(self.buffer_size):
            if self._lib.gvom_slot_filled(self._h, i) != 1:
                out.append(None)
            else:
                out.append(_DeviceArrayView(lambda i=i: self.read_dense(i)[4]))
        return out

    def _fused_view(self, field):
        if not self._state().has_combined:
            return None
        return _DeviceArrayView(lambda: self._compact(GVOM_WHICH_FUSED)[field])

    combined_index_map = property(lambda self: self._fused_view(0))
    combined_hit_count = property(lambda self: self._fused_view(1))
    combined_total_count = property(lambda self: self._fused_view(2))
    combined_min_height = property(lambda self: self._fused_view(3))
    last_combined_index_map = combined_index_map
    last_combined_hit_count = combined_hit_count
    last_combined_total_count = combined_total_count
    last_combined_min_height = combined_min_height

    @property
    def combined_origin(self):
        st = self._state()
        if not st.has_combined:
            return None
        org = np.array(list(st.combined_origin), np.float64)
        return _DeviceArrayView(lambda: org.copy())

    last_combined_origin = combined_origin

    def _map2d(self, which):
        out = np.empty((self.xy_size, self.xy_size), np.float64)
        rc = self._check(self._lib.gvom_read_map2d(self._h, which, _ptr(out)))
        return None if rc == GVOM_NO_DATA else out

    def _map_view(self, which):
        if self._map2d(which) is None:
            return None
        return _DeviceArrayView(lambda: self._map2d(which))

    height_map = property(lambda self: self._map_view(MAP_HEIGHT))
    inferred_height_map = property(lambda self: self._map_view(MAP_INFERRED))
    x_slope_map = property(lambda self: self._map_view(MAP_SLOPE_X))
    y_slope_map = property(lambda self: self._map_view(MAP_SLOPE_Y))
    roughness_map = property(lambda self: self._map_view(MAP_ROUGHNESS))
    guessed_height_delta = property(lambda self: self._map_view(MAP_GUESSED))

    # ---- measurement helpers ----------------------------------------------------------
    def set_profiling(self, on):
        self._check(self._lib.gvom_set_profiling(self._h, 1 if on else 0))

    def last_stage_ms(self):
        ms = (ctypes.c_float * N_STAGES)()
        self._check(self._lib.gvom_last_stage_ms(self._h, ctypes.byref(ms)))
        return dict(zip(STAGE_NAMES, [float(v) for v in ms]))

    def set_tuning(self, name, value):
        """Performance knobs that never change a result: "segs", "period", "ep_row", "prio", "interleave", "eager" (include/gvom_hip.h)."""
        self._check(self._lib.gvom_set_tuning(self._h, name.encode(), int(value)))

    def get_tuning(self, name):
        """The value the last scan ran with (what "automatic" resolved to)."""
        v = _I(0)
        self._check(self._lib.gvom_get_tuning(self._h, name.encode(), ctypes.byref(v)))
        return int(v.value)

    def host_timing(self):
        us = (ctypes.c_double * 8)()
        self._check(self._lib.gvom_host_timing(self._h, ctypes.byref(us)))
        return dict(zip(("scan_launch", "scan_wait", "combine_launch", "combine_wait", "output_copy"),
                        [float(v) for v in us][:5]))

    def scan_stats(self):
        st = GvomScanStats()
        rc = self._check(self._lib.gvom_get_scan_stats(self._h, ctypes.byref(st)))
        if rc == GVOM_NO_DATA:
            return None
        return {"points": int(st.points), "cells": int(st.cells), "sum_hit": int(st.sum_hit),
                "sum_total": int(st.sum_total)}

    @staticmethod
    def backend_info():
        L = load_library()
        buf = ctypes.create_string_buffer(256)
        rc = L.gvom_backend_info(buf, 256)
        return rc, buf.value.decode()

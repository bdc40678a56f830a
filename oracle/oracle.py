"""ctypes front-end of the CPU oracle (oracle/gvom_oracle.c).

TEST INFRASTRUCTURE ONLY -- imported by tests/, __graft_entry__.smoke() and the
`cpu_baseline` leg of bench.py; never by the product package g-vom_amd/.

`OracleGvom` restates the *host* side of the reference class
(/root/reference/scripts/gvom.py:29-354, "gvom.py:NNN" below) on numpy arrays, calling
the C restatement of each kernel in the reference's launch order.  It keeps the
reference's attribute names (index_buffer, hit_count_buffer, combined_index_map,
height_map, ...) so that golden vectors captured from the reference map one-to-one.
"""
import ctypes
import math
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# (GVOM_ORACLE_LIBRARY / GVOM_ORACLE_OMP_LIBRARY: the sanitizer builds of `make -C oracle san`, tests/test_sanitizers.py)
_LIB_PATH = os.environ.get("GVOM_ORACLE_LIBRARY", os.path.join(_HERE, "libgvom_oracle.so"))
_OMP_PATH = os.environ.get("GVOM_ORACLE_OMP_LIBRARY", os.path.join(_HERE, "libgvom_oracle_omp.so"))   # the same source on all host cores (-fopenmp -DORC_OMP)

_c = ctypes
_i64, _f64 = _c.c_int64, _c.c_double
_P = _c.c_void_p


def build(force=False):
    """Compile the oracle with gcc (oracle/Makefile)."""
    src = os.path.join(_HERE, "gvom_oracle.c")
    for path in (_LIB_PATH, _OMP_PATH):
        if os.path.dirname(os.path.abspath(path)) != _HERE:
            continue                                   # a sanitizer build handed in through the environment: built by its own recipe
        if force or not os.path.exists(path) or os.path.getmtime(path) < os.path.getmtime(src):
            subprocess.check_call(["make", "-C", _HERE, "-B", os.path.basename(path)], stdout=subprocess.DEVNULL)
    return _LIB_PATH


_lib = None
_libs = {}


def use_all_cores(on=True, threads=None):
    """Switches every oracle call of this process to the all-core (OpenMP) build of the same source,
    or back to the one-thread build.  Returns the number of threads in use."""
    global _lib
    _lib = None
    L = lib(_OMP_PATH if on else _LIB_PATH)
    if on and threads:
        L.orc_set_threads(int(threads))
    return int(L.orc_threads())


def lib(path=None):
    global _lib
    if _lib is None or path is not None:
        build()
        path = path or _LIB_PATH
        if path in _libs:
            _lib = _libs[path]
            return _lib
        L = ctypes.CDLL(path)
        L.orc_threads.restype = _c.c_int
        L.orc_set_cuda_f32_sqrt.argtypes = [_c.c_int]
        L.orc_set_threads.argtypes = [_c.c_int]
        L.orc_fill_i32.argtypes = [_P, _i64, _c.c_int32]; L.orc_fill_i32.restype = None
        for suf in ("f32", "f64"):
            f = getattr(L, "orc_transform_pointcloud_" + suf)
            f.argtypes = [_P, _i64, _i64, _P]; f.restype = None
            f = getattr(L, "orc_point_2_map_" + suf)
            f.argtypes = [_f64, _f64, _i64, _i64, _f64, _P, _i64, _i64, _P, _P, _P, _P]
            f.restype = _i64
            f = getattr(L, "orc_calculate_min_height_" + suf)
            f.argtypes = [_f64, _f64, _i64, _i64, _f64, _P, _P, _i64, _i64, _P, _P]
            f.restype = _i64
            f = getattr(L, "orc_calculate_stats_" + suf)
            f.argtypes = [_c.c_int, _f64, _f64, _i64, _i64, _f64, _P, _P, _i64, _i64, _P, _P, _i64, _i64]
            f.restype = None
            f = getattr(L, "orc_combine_metrics_full_" + suf)
            f.argtypes = [_P] * 12 + [_i64, _i64]; f.restype = None
        L.orc_normalize_stats.argtypes = [_c.c_int, _P, _i64]; L.orc_normalize_stats.restype = None
        L.orc_calculate_eigenvalues.argtypes = [_P, _P, _i64]; L.orc_calculate_eigenvalues.restype = None
        L.orc_make_voxel_pointcloud.argtypes = [_P, _P, _P, _P, _P, _P, _i64, _i64, _f64, _f64]
        L.orc_make_voxel_pointcloud.restype = None
        L.orc_assign_indices.argtypes = [_P, _P, _P, _i64]; L.orc_assign_indices.restype = _c.c_int32
        L.orc_move_data.argtypes = [_P, _P, _P, _i64]; L.orc_move_data.restype = None
        for name in ("orc_combine_indices", "orc_combine_old_indices"):
            f = getattr(L, name); f.argtypes = [_P, _P, _P, _P, _P, _i64, _i64]; f.restype = None
        L.orc_combine_metrics.argtypes = [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _i64, _i64]
        L.orc_combine_metrics.restype = None
        L.orc_make_height_map.argtypes = [_P, _P, _P, _i64, _i64, _f64, _f64, _P, _f64, _f64, _P]
        L.orc_make_height_map.restype = None
        L.orc_make_inferred_height_map.argtypes = [_P, _P, _i64, _i64, _f64, _P]
        L.orc_make_inferred_height_map.restype = None
        L.orc_calculate_slope.argtypes = [_P, _i64, _f64, _P, _P, _P]
        L.orc_calculate_slope.restype = None
        L.orc_guess_height.argtypes = [_P, _P, _i64, _P]; L.orc_guess_height.restype = None
        L.orc_make_positive_obstacle_map.argtypes = [_P, _P, _i64, _i64, _f64, _f64, _P, _P, _f64,
                                                     _P, _P, _P, _f64, _P]
        L.orc_make_positive_obstacle_map.restype = None
        L.orc_make_negative_obstacle_map.argtypes = [_P, _P, _f64, _i64]
        L.orc_make_negative_obstacle_map.restype = None
        L.orc_make_visibility_map.argtypes = [_P, _P, _i64]; L.orc_make_visibility_map.restype = None
        L.orc_make_height_map_pointcloud.argtypes = [_P, _P, _P, _P, _P, _P, _i64, _f64, _f64]
        L.orc_make_height_map_pointcloud.restype = None
        L.orc_make_inferred_height_map_pointcloud.argtypes = [_P, _P, _P, _i64, _f64, _f64]
        L.orc_make_inferred_height_map_pointcloud.restype = None
        _libs[path] = L
        _lib = L
    return _lib


def _p(a):
    return a.ctypes.data_as(_P)


def _as_cloud(pointcloud):
    """(N, >=3) float32/float64 C-contiguous copy (the reference copies H2D, gvom.py:110)."""
    pc = np.asarray(pointcloud)
    if pc.dtype not in (np.float32, np.float64):
        pc = pc.astype(np.float64)
    return np.array(pc, order="C", copy=True)


# ---- kernel-level entry points (used by known-answer tests) -------------------------

def point_2_map(xy_res, z_res, xy, zs, min_distance, points, ego, origin):
    """gvom.py:1060-1150 on a fresh grid. Returns (hit[V], total[V], n_updates)."""
    pc = _as_cloud(points)
    suf = "f32" if pc.dtype == np.float32 else "f64"
    V = xy * xy * zs
    hit = np.zeros(V, np.int32); total = np.zeros(V, np.int32)
    ego = np.asarray(ego, np.float64); origin = np.asarray(origin, np.float64)
    n = getattr(lib(), "orc_point_2_map_" + suf)(xy_res, z_res, xy, zs, min_distance, _p(pc),
                                               pc.shape[0], pc.shape[1], _p(hit), _p(total),
                                               _p(ego), _p(origin))
    return hit, total, n


def transform_pointcloud(points, transform):
    pc = _as_cloud(points)
    suf = "f32" if pc.dtype == np.float32 else "f64"
    tf = np.ascontiguousarray(np.asarray(transform, np.float64))
    getattr(lib(), "orc_transform_pointcloud_" + suf)(_p(pc), pc.shape[0], pc.shape[1], _p(tf))
    return pc


def calculate_slope(height_map, xy_res):
    h = np.ascontiguousarray(height_map, np.float64); xy = h.shape[0]
    sx = np.zeros((xy, xy)); sy = np.zeros((xy, xy)); r = np.full((xy, xy), -1.0)
    lib().orc_calculate_slope(_p(h), xy, xy_res, _p(sx), _p(sy), _p(r))
    return sx, sy, r


def guess_height(height_map, inferred_height_map):
    h = np.ascontiguousarray(height_map, np.float64); xy = h.shape[0]
    inf = np.ascontiguousarray(inferred_height_map, np.float64)
    out = np.zeros((xy, xy))
    lib().orc_guess_height(_p(h), _p(inf), xy, _p(out))
    return out


def make_positive_obstacle_map(index_map, height_map, xy, zs, z_res, pos_thr, hit, total,
                               robot_height, origin, x_slope, y_slope, slope_thr):
    out = np.zeros((xy, xy), np.int32)
    args = [np.ascontiguousarray(index_map, np.int32), np.ascontiguousarray(height_map, np.float64),
            np.ascontiguousarray(hit, np.int32), np.ascontiguousarray(total, np.int32),
            np.asarray(origin, np.float64), np.ascontiguousarray(x_slope, np.float64),
            np.ascontiguousarray(y_slope, np.float64)]
    lib().orc_make_positive_obstacle_map(_p(args[0]), _p(args[1]), xy, zs, z_res, pos_thr,
                                         _p(args[2]), _p(args[3]), robot_height, _p(args[4]),
                                         _p(args[5]), _p(args[6]), slope_thr, _p(out))
    return out


def combine_old_indices(combined_index_map, old_index_map, xy, zs, combined_origin=(0, 0, 0),
                        old_origin=(0, 0, 0), cell_count=0):
    c = np.array(combined_index_map, np.int32).copy(); o = np.ascontiguousarray(old_index_map, np.int32)
    cnt = np.array([cell_count], np.int64)
    co = np.asarray(combined_origin, np.float64); oo = np.asarray(old_origin, np.float64)
    lib().orc_combine_old_indices(_p(cnt), _p(c), _p(co), _p(o), _p(oo), xy, zs)
    return c, int(cnt[0])


# ---- the class ------------------------------------------------------------------------

class OracleGvom:
    """Host-side restatement of reference class Gvom (gvom.py:12-410), numpy + C oracle."""

    def __init__(self, xy_resolution, z_resolution, xy_size, z_size, buffer_size, min_distance,
                 positive_obstacle_threshold, negative_obstacle_threshold, slope_obstacle_threshold,
                 robot_height, robot_radius, ground_to_lidar_height, xy_eigen_dist, z_eigen_dist,
                 voxel_statistics=False, cuda_f32_sqrt=False, numba_cuda_typing=None, c_order=False):
        # cuda_f32_sqrt: ray_length = sqrt(float32) evaluated in float32, as Numba types gvom.py:1109 for a
        # real CUDA device; default = the simulator's float64 square root (the fixtures)
        self.cuda_f32_sqrt = bool(cuda_f32_sqrt if numba_cuda_typing is None else numba_cuda_typing)   # (numba_cuda_typing: the switch's name since round 6)
        # voxel_statistics: also restate the per-voxel mean/covariance/eigenvalue path (SURVEY 8f
        # rank 2: gvom.py:1172-1299, 858-909, 1333-1378, 363-378).  Off by default so that the
        # timed CPU baseline covers the same work as the GPU hot path.
        self.voxel_statistics = voxel_statistics
        self.metrics_buffer = [None] * buffer_size
        self.combined_metrics = None
        self.last_combined_metrics = None
        self.voxels_eigenvalues = None
        self.xy_resolution = xy_resolution
        self.z_resolution = z_resolution
        self.xy_size = xy_size
        self.z_size = z_size
        self.voxel_count = xy_size * xy_size * z_size
        self.min_distance = min_distance
        self.positive_obstacle_threshold = positive_obstacle_threshold
        self.negative_obstacle_threshold = negative_obstacle_threshold
        self.slope_obstacle_threshold = slope_obstacle_threshold
        self.robot_height = robot_height
        self.robot_radius = robot_radius
        self.ground_to_lidar_height = ground_to_lidar_height
        self.xy_eigen_dist = xy_eigen_dist
        self.z_eigen_dist = z_eigen_dist
        self.buffer_size = buffer_size
        self.buffer_index = 0
        self.last_buffer_index = 0
        self.index_buffer = [None] * buffer_size
        self.hit_count_buffer = [None] * buffer_size
        self.total_count_buffer = [None] * buffer_size
        self.origin_buffer = [None] * buffer_size
        self.min_height_buffer = [None] * buffer_size
        self.combined_index_map = None
        self.combined_hit_count = None
        self.combined_total_count = None
        self.combined_min_height = None
        self.combined_origin = None
        self.combined_cell_count_cpu = None
        self.last_combined_index_map = None
        self.last_combined_hit_count = None
        self.last_combined_total_count = None
        self.last_combined_min_height = None
        self.last_combined_origin = None
        self.height_map = None
        self.inferred_height_map = None
        self.roughness_map = None
        self.guessed_height_delta = None
        self.x_slope_map = None
        self.y_slope_map = None
        self.ego_position = [0, 0, 0]
        # accounting for bench.py's roofline arithmetic (exact integers)
        self.last_scan_updates = 0
        self.last_scan_points_in_grid = 0
        # timed baseline only (bench.py cpu_baseline): V-sized int32 arrays that have left the ring / been replaced are filled
        # and used again instead of being given back to the allocator (fresh pages cost a page fault per 4 KiB, in ONE
        # thread for np.full: most of an all-core step).  Off by default: arrays handed out as attributes stay untouched.
        self.reuse_buffers = False
        self._free_v = []

    def _take_v(self, value):
        """a V-sized int32 array filled with `value` (gvom.py:114-121, 190: cuda.device_array + __init_1D_array), filled by the C
        oracle -- on all threads in its all-core build"""
        a = self._free_v.pop() if self._free_v else np.empty(self.voxel_count, np.int32)
        lib().orc_fill_i32(_p(a), self.voxel_count, int(value))
        return a

    def _give_v(self, a):
        if self.reuse_buffers and a is not None and len(self._free_v) < 8:
            self._free_v.append(a)

    # gvom.py:99-175
    def process_pointcloud(self, pointcloud, ego_position, transform=None):
        L = lib()
        self.ego_position = ego_position
        point_count = pointcloud.shape[0]
        if point_count == 0:
            print("[WARNING] Processing an empty pointcloud, nothing will happen!")
            return
        pc = _as_cloud(pointcloud)
        suf = "f32" if pc.dtype == np.float32 else "f64"
        V = self.voxel_count
        tmp_hit = self._take_v(0)
        tmp_total = self._take_v(0)
        index_map = self._take_v(-1)
        origin = np.zeros(3)
        origin[0] = math.floor((ego_position[0] / self.xy_resolution) - self.xy_size / 2)   # :124
        origin[1] = math.floor((ego_position[1] / self.xy_resolution) - self.xy_size / 2)
        origin[2] = math.floor((ego_position[2] / self.z_resolution) - self.z_size / 2)
        ego = np.asarray(ego_position, dtype=np.float64)
        if transform is not None:                                                            # :134
            tf = np.ascontiguousarray(np.asarray(transform, np.float64))
            getattr(L, "orc_transform_pointcloud_" + suf)(_p(pc), point_count, pc.shape[1], _p(tf))
        L.orc_set_cuda_f32_sqrt(1 if self.cuda_f32_sqrt else 0)
        self.last_scan_updates = getattr(L, "orc_point_2_map_" + suf)(                      # :138
            self.xy_resolution, self.z_resolution, self.xy_size, self.z_size, self.min_distance,
            _p(pc), point_count, pc.shape[1], _p(tmp_hit), _p(tmp_total), _p(ego), _p(origin))
        cell_count = L.orc_assign_indices(_p(tmp_hit), _p(tmp_total), _p(index_map), V)      # :143
        if cell_count == 0:                                                                  # :148
            print("[WARNING] The pointcloud points don't overlap with any voxels, nothing will happen!")
            self._give_v(tmp_hit); self._give_v(tmp_total); self._give_v(index_map)
            return
        hit = np.empty(cell_count, np.int32); total = np.empty(cell_count, np.int32)
        L.orc_move_data(_p(tmp_hit), _p(hit), _p(index_map), V)                              # :155
        L.orc_move_data(_p(tmp_total), _p(total), _p(index_map), V)
        min_height = np.ones(cell_count * 3, np.float32)                                     # :1014
        self.last_scan_points_in_grid = getattr(L, "orc_calculate_min_height_" + suf)(      # :1032
            self.xy_resolution, self.z_resolution, self.xy_size, self.z_size, self.min_distance,
            _p(index_map), _p(pc), point_count, pc.shape[1], _p(min_height), _p(origin))
        metrics = None
        if self.voxel_statistics:                                                            # :1011-1030
            metrics = np.zeros((cell_count, 10), np.float64)
            for ps in (0, 1):
                getattr(L, "orc_calculate_stats_" + suf)(
                    ps, self.xy_resolution, self.z_resolution, self.xy_size, self.z_size,
                    self.min_distance, _p(index_map), _p(pc), point_count, pc.shape[1], _p(metrics),
                    _p(origin), self.xy_eigen_dist, self.z_eigen_dist)
                L.orc_normalize_stats(ps, _p(metrics), cell_count)
        self._give_v(tmp_hit); self._give_v(tmp_total)
        b = self.buffer_index                                                                # :163
        self._give_v(self.index_buffer[b])
        self.metrics_buffer[b] = metrics
        self.index_buffer[b] = index_map
        self.hit_count_buffer[b] = hit
        self.total_count_buffer[b] = total
        self.min_height_buffer[b] = min_height
        self.origin_buffer[b] = origin
        self.last_buffer_index = b
        self.buffer_index += 1
        if self.buffer_index >= self.buffer_size:
            self.buffer_index = 0

    # gvom.py:177-354
    def combine_maps(self):
        L = lib()
        if self.origin_buffer[self.last_buffer_index] is None:
            print("[WARNING] The map buffer is empty, nothing will happen!")
            return None
        xy, zs, V = self.xy_size, self.z_size, self.voxel_count
        self.combined_origin = self.origin_buffer[self.last_buffer_index].copy()             # :184
        origin_world = self.combined_origin.copy()
        origin_world[0] = origin_world[0] * self.xy_resolution
        origin_world[1] = origin_world[1] * self.xy_resolution
        origin_world[2] = origin_world[2] * self.z_resolution
        cnt = np.zeros(1, np.int64)
        self.combined_index_map = self._take_v(-1)
        for i in range(self.buffer_size):                                                    # :198
            if self.origin_buffer[i] is None:
                continue
            L.orc_combine_indices(_p(cnt), _p(self.combined_index_map), _p(self.combined_origin),
                                  _p(self.index_buffer[i]), _p(self.origin_buffer[i]), xy, zs)
        if self.last_combined_origin is not None:                                            # :210
            L.orc_combine_old_indices(_p(cnt), _p(self.combined_index_map), _p(self.combined_origin),
                                      _p(self.last_combined_index_map),
                                      _p(self.last_combined_origin), xy, zs)
        Cc = int(cnt[0])
        self.combined_cell_count_cpu = Cc
        self.combined_hit_count = np.zeros(Cc, np.int32)
        self.combined_total_count = np.zeros(Cc, np.int32)
        self.combined_min_height = np.ones(Cc, np.float32)
        if self.voxel_statistics:
            self._combine_with_statistics(Cc)
        for i in range(self.buffer_size if not self.voxel_statistics else 0):                # :238
            if self.origin_buffer[i] is None:
                continue
            L.orc_combine_metrics(_p(self.combined_hit_count), _p(self.combined_total_count),
                                  _p(self.combined_min_height), _p(self.combined_index_map),
                                  _p(self.combined_origin), _p(self.hit_count_buffer[i]),
                                  _p(self.total_count_buffer[i]), _p(self.min_height_buffer[i]),
                                  _p(self.index_buffer[i]), _p(self.origin_buffer[i]), xy, zs)
        if self.last_combined_origin is not None and not self.voxel_statistics:              # :254
            L.orc_combine_metrics(_p(self.combined_hit_count), _p(self.combined_total_count),
                                  _p(self.combined_min_height), _p(self.combined_index_map),
                                  _p(self.combined_origin), _p(self.last_combined_hit_count),
                                  _p(self.last_combined_total_count),
                                  _p(self.last_combined_min_height),
                                  _p(self.last_combined_index_map), _p(self.last_combined_origin),
                                  xy, zs)
        self.last_combined_hit_count = self.combined_hit_count                               # :268
        self.last_combined_total_count = self.combined_total_count
        self._give_v(self.last_combined_index_map)
        self.last_combined_index_map = self.combined_index_map
        self.last_combined_min_height = self.combined_min_height
        self.last_combined_origin = self.combined_origin

        self.height_map = np.full((xy, xy), -1000.0)                                         # :288
        self.inferred_height_map = np.full((xy, xy), -1000.0)
        ego = np.asarray(self.ego_position, dtype=np.float64)
        L.orc_make_height_map(_p(self.combined_origin), _p(self.combined_index_map),
                              _p(self.combined_min_height), xy, zs, self.xy_resolution,
                              self.z_resolution, _p(ego), self.robot_radius,
                              self.ground_to_lidar_height, _p(self.height_map))
        L.orc_make_inferred_height_map(_p(self.combined_origin), _p(self.combined_index_map), xy, zs,
                                       self.z_resolution, _p(self.inferred_height_map))
        self.roughness_map = np.full((xy, xy), -1.0)                                         # :307
        self.x_slope_map = np.zeros((xy, xy))
        self.y_slope_map = np.zeros((xy, xy))
        L.orc_calculate_slope(_p(self.height_map), xy, self.xy_resolution, _p(self.x_slope_map),
                              _p(self.y_slope_map), _p(self.roughness_map))
        self.guessed_height_delta = np.zeros((xy, xy))                                       # :320
        L.orc_guess_height(_p(self.height_map), _p(self.inferred_height_map), xy,
                           _p(self.guessed_height_delta))
        positive = np.zeros((xy, xy), np.int32)                                              # :329
        L.orc_make_positive_obstacle_map(_p(self.combined_index_map), _p(self.height_map), xy, zs,
                                         self.z_resolution, self.positive_obstacle_threshold,
                                         _p(self.combined_hit_count), _p(self.combined_total_count),
                                         self.robot_height, _p(self.combined_origin),
                                         _p(self.x_slope_map), _p(self.y_slope_map),
                                         self.slope_obstacle_threshold, _p(positive))
        negative = np.zeros((xy, xy), np.int32)                                              # :341
        L.orc_make_negative_obstacle_map(_p(self.guessed_height_delta), _p(negative),
                                         self.negative_obstacle_threshold, xy)
        visibility = np.zeros((xy, xy), np.int32)                                            # :348
        L.orc_make_visibility_map(_p(visibility), _p(self.height_map), xy)
        return (origin_world, positive, negative, self.roughness_map.copy(), visibility)

    def _combine_with_statistics(self, Cc):
        """gvom.py:234-284 with the covariance merge: slots (float64 metrics) in slot order, then
        the previous fused map (float32 metrics), then the eigenvalues."""
        L = lib()
        xy, zs = self.xy_size, self.z_size
        self.combined_metrics = np.zeros((Cc, 10), np.float32)
        for i in range(self.buffer_size):
            if self.origin_buffer[i] is None:
                continue
            L.orc_combine_metrics_full_f64(
                _p(self.combined_metrics), _p(self.combined_hit_count), _p(self.combined_total_count),
                _p(self.combined_min_height), _p(self.combined_index_map), _p(self.combined_origin),
                _p(self.metrics_buffer[i]), _p(self.hit_count_buffer[i]), _p(self.total_count_buffer[i]),
                _p(self.min_height_buffer[i]), _p(self.index_buffer[i]), _p(self.origin_buffer[i]), xy, zs)
        if self.last_combined_origin is not None:
            L.orc_combine_metrics_full_f32(
                _p(self.combined_metrics), _p(self.combined_hit_count), _p(self.combined_total_count),
                _p(self.combined_min_height), _p(self.combined_index_map), _p(self.combined_origin),
                _p(self.last_combined_metrics), _p(self.last_combined_hit_count),
                _p(self.last_combined_total_count), _p(self.last_combined_min_height),
                _p(self.last_combined_index_map), _p(self.last_combined_origin), xy, zs)
        self.last_combined_metrics = self.combined_metrics
        self.voxels_eigenvalues = np.zeros((Cc, 3), np.float32)
        L.orc_calculate_eigenvalues(_p(self.voxels_eigenvalues), _p(self.combined_metrics), Cc)

    # gvom.py:363-378
    def make_debug_voxel_map(self):
        if self.combined_cell_count_cpu is None:
            print("No data")
            return None
        if not self.voxel_statistics:
            return None
        out = np.zeros([self.combined_cell_count_cpu, 8], np.float32)
        lib().orc_make_voxel_pointcloud(_p(self.combined_index_map), _p(self.combined_hit_count),
                                        _p(self.combined_total_count), _p(self.voxels_eigenvalues),
                                        _p(self.combined_origin), _p(out), self.xy_size, self.z_size,
                                        self.xy_resolution, self.z_resolution)
        return out

    # gvom.py:356-361
    def get_map_as_occupancy_grid(self):
        lut = self.last_combined_index_map.reshape((self.xy_size, self.xy_size, self.z_size), order="F")
        return lut >= 0

    # gvom.py:380-394
    def make_debug_height_map(self):
        if self.height_map is None:
            print("No data")
            return None
        out = np.zeros([self.xy_size * self.xy_size, 7], np.float32)
        lib().orc_make_height_map_pointcloud(_p(self.height_map), _p(self.roughness_map),
                                             _p(self.x_slope_map), _p(self.y_slope_map),
                                             _p(self.combined_origin), _p(out), self.xy_size,
                                             self.xy_resolution, self.z_resolution)
        return out

    # gvom.py:396-410
    def make_debug_inferred_height_map(self):
        if self.height_map is None:
            print("No data")
            return None
        out = np.zeros([self.xy_size * self.xy_size, 3], np.float32)
        lib().orc_make_inferred_height_map_pointcloud(_p(self.guessed_height_delta),
                                                      _p(self.combined_origin), _p(out),
                                                      self.xy_size, self.xy_resolution,
                                                      self.z_resolution)
        return out


# ---- dense equivalents (what parity tests compare; compact row order is unspecified) ----

def dense_from_compact(index_map, hit, total, min_height):
    """(state, hit_dense, total_dense, min_h_dense) with state = index_map where rows >= 0
    are canonicalised to 0; free/unknown codes (< 0) kept; min_h_dense = 1.0 where empty."""
    index_map = np.asarray(index_map)
    occ = index_map >= 0
    state = np.where(occ, 0, index_map).astype(np.int32)
    hd = np.zeros(index_map.shape, np.int32); td = np.zeros(index_map.shape, np.int32)
    md = np.ones(index_map.shape, np.float32)
    rows = index_map[occ]
    hd[occ] = np.asarray(hit)[rows]; td[occ] = np.asarray(total)[rows]
    md[occ] = np.asarray(min_height)[rows]
    return state, hd, td, md


def ros_occupancy_grids(map_data, density_threshold=50, min_roughness=-10, max_roughness=0):
    """The post-processing the ROS node applies to combine_maps()'s result before publishing
    (reference gvom_ros.py:141-165), restated with the same numpy operations:
    returns the five int8 nav_msgs/OccupancyGrid.data arrays (hard, soft, certainty, negative,
    roughness).  Parameter defaults = the node's ROS parameter defaults (gvom_ros.py:32-35).
    As written, the roughness rescale ADDS min_roughness and the int8 cast wraps."""
    obs_map, neg_map, rough_map, cert_map = map_data[1], map_data[2], map_data[3], map_data[4]
    with np.errstate(invalid="ignore"):
        hard = np.reshape(np.maximum(100 * (obs_map > density_threshold), neg_map), -1, order='F').astype(np.int8)   # :141
        soft = np.reshape(100 * (obs_map <= density_threshold) * (obs_map > 0), -1, order='F').astype(np.int8)       # :146
        cert = np.reshape(cert_map * 100, -1, order='F').astype(np.int8)                                               # :151
        neg = np.reshape(neg_map, -1, order='F').astype(np.int8)                                                       # :157
        r = ((np.maximum(np.minimum(rough_map, max_roughness), min_roughness) + min_roughness)
             / (max_roughness - min_roughness)) * 100                                                                  # :162
        rough = np.reshape(r, -1, order='F').astype(np.int8)                                                           # :163
    return hard, soft, cert, neg, rough


def pointcloud2_to_xyz_array(data, n_points, point_step, offsets, field_dtype=np.float32, remove_nans=True):
    """What the node feeds process_pointcloud (reference gvom_ros.py:108): ros_numpy 0.0.x
    point_cloud2.pointcloud2_to_xyz_array = get_xyz_points(pointcloud2_to_array(msg)) -- a float64
    (dtype=np.float) [N', 3] array of the x, y, z fields with the records that have a non-finite
    coordinate removed.  ros_numpy is not part of the reference checkout; restated from its
    published source."""
    rec = np.dtype({"names": ["x", "y", "z"], "formats": [np.dtype(field_dtype)] * 3,
                    "offsets": [int(o) for o in offsets], "itemsize": int(point_step)})
    arr = np.frombuffer(data, dtype=rec, count=int(n_points))
    if remove_nans:
        mask = np.isfinite(arr["x"]) & np.isfinite(arr["y"]) & np.isfinite(arr["z"])
        arr = arr[mask]
    pts = np.zeros(arr.shape + (3,), dtype=np.float64)
    pts[..., 0] = arr["x"]; pts[..., 1] = arr["y"]; pts[..., 2] = arr["z"]
    return pts

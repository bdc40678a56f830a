"""Test double for the per-rank compute of g-vom_amd/gvom_sharded.py: the CPU oracle stands in
for the HIP library so that the SHARDING LOGIC (cloud all-gather, global cell count, height
row all-gather in storage order, output reassembly) runs under gloo on CPU.

Each rank computes with the full oracle but only ever EXPORTS the rows of its own slab
(everything else is poisoned), and derives its 2-D outputs from the IMPORTED (gathered)
height maps -- so a wrong row order / wrong window<->storage conversion / missing gather
changes the final maps."""
import contextlib
import io

import numpy as np
import torch

import gvom as _gvom
from oracle import oracle

POISON = -7777.0


class OracleShardBackend(object):
    def __init__(self, params, rank, world):
        self.g = oracle.OracleGvom(*params)
        self.params = params
        self.rank, self.world = rank, world
        self.xy, self.zs = params[2], params[3]
        self.rows = self.xy // world
        self.lo, self.hi = rank * self.rows, (rank + 1) * self.rows
        self.maps = {}

    # -- layout helpers ---------------------------------------------------------------------
    def _om(self, origin):
        return int(origin[0]) % self.xy, int(origin[1]) % self.xy

    def _to_storage(self, m_xy, origin):          # reference [x][y] -> storage [sy][sx]
        om0, om1 = self._om(origin)
        return np.roll(np.ascontiguousarray(m_xy.T), (om1, om0), (0, 1))

    def _to_window(self, s_yx, origin):           # storage [sy][sx] -> reference [x][y]
        om0, om1 = self._om(origin)
        return np.ascontiguousarray(np.roll(s_yx, (-om1, -om0), (0, 1)).T)

    def _own_cells(self, index_map, origin):
        occ = (np.asarray(index_map) >= 0).reshape(self.zs, self.xy, self.xy)   # [z][y][x]
        om1 = int(origin[1]) % self.xy
        sy = (np.arange(self.xy) + om1) % self.xy
        own = (sy >= self.lo) & (sy < self.hi)
        return int(occ[:, own, :].sum())

    # -- interface used by ShardedGvom ----------------------------------------------------------
    def empty_rows(self, which, full=False):
        dt = torch.float64 if which in (_gvom.MAP_HEIGHT, _gvom.MAP_INFERRED, _gvom.OUT_ROUGHNESS) else torch.int32
        return torch.empty(((self.xy if full else self.rows), self.xy), dtype=dt)

    def cloud_tensor(self, pc):
        return torch.from_numpy(np.ascontiguousarray(pc[:, :3]))

    def scan_begin(self, cloud, ego, tf):
        pc = cloud.numpy()
        if pc.shape[0] == 0:
            self.g.ego_position = ego
            return _gvom.GVOM_EMPTY_CLOUD, 0
        slot = self.g.buffer_index
        was = self.g.origin_buffer[slot]
        with contextlib.redirect_stdout(io.StringIO()):
            self.g.process_pointcloud(pc, ego, tf)
        if self.g.origin_buffer[slot] is was:          # rejected by the oracle's own global test
            return _gvom.GVOM_OK, 0
        return _gvom.GVOM_OK, self._local_scan_cells(slot)

    def _local_scan_cells(self, slot):
        if self.g.origin_buffer[slot] is None:
            return 0
        return self._own_cells(self.g.index_buffer[slot], self.g.origin_buffer[slot])

    def scan_commit(self, accept):
        pass                                       # the oracle applied the same global rule itself

    def combine_fuse(self):
        with contextlib.redirect_stdout(io.StringIO()):
            out = self.g.combine_maps()
        if out is None:
            return _gvom.GVOM_EMPTY_BUFFER, 0
        self.origin = self.g.combined_origin
        for which, m in ((_gvom.MAP_HEIGHT, self.g.height_map), (_gvom.MAP_INFERRED, self.g.inferred_height_map)):
            s = self._to_storage(m, self.origin)
            s[:self.lo] = POISON; s[self.hi:] = POISON          # only own rows are "computed"
            self.maps[which] = s
        return _gvom.GVOM_OK, self._own_cells(self.g.combined_index_map, self.origin)

    def set_cell_count(self, n):
        self.cell_count = n

    def rows_export(self, which):
        return torch.from_numpy(np.ascontiguousarray(self.maps[which][self.lo:self.hi]))

    def rows_import(self, which, full):
        self.maps[which] = full.numpy().copy()

    def combine_map2d(self):
        g = self.g
        h = self._to_window(self.maps[_gvom.MAP_HEIGHT], self.origin)
        inf = self._to_window(self.maps[_gvom.MAP_INFERRED], self.origin)
        sx, sy, r = oracle.calculate_slope(h, g.xy_resolution)
        dh = oracle.guess_height(h, inf)
        pos = oracle.make_positive_obstacle_map(g.combined_index_map, h, self.xy, self.zs, g.z_resolution,
                                                g.positive_obstacle_threshold, g.combined_hit_count,
                                                g.combined_total_count, g.robot_height, self.origin,
                                                sx, sy, g.slope_obstacle_threshold)
        neg = np.where(dh > g.negative_obstacle_threshold, 100, 0).astype(np.int32)
        vis = (h > -1000).astype(np.int32)
        for which, m in ((_gvom.OUT_POSITIVE, pos), (_gvom.OUT_NEGATIVE, neg), (_gvom.OUT_ROUGHNESS, r),
                         (_gvom.OUT_VISIBILITY, vis)):
            s = self._to_storage(m, self.origin)
            s[:self.lo] = -7777; s[self.hi:] = -7777
            self.maps[which] = s

    def finalize(self):
        o = self.origin.copy()
        o[0] *= self.g.xy_resolution; o[1] *= self.g.xy_resolution; o[2] *= self.g.z_resolution
        return (o,) + tuple(self._to_window(self.maps[w], self.origin)
                            for w in (_gvom.OUT_POSITIVE, _gvom.OUT_NEGATIVE, _gvom.OUT_ROUGHNESS,
                                      _gvom.OUT_VISIBILITY))

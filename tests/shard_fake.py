"""Test double for the per-rank compute of g-vom_amd/gvom_sharded.py: the CPU oracle stands in
for the HIP library so that the SHARDING LOGIC (cloud all-gather, asynchronous global cell
count + lazy commit, in-place all-gather of the row-interleaved height buffer and of the packed
output rows, storage<->window reordering) runs under gloo on CPU.

Each rank computes with the full oracle but only ever PUBLISHES the rows of its own slab
(everything else is poisoned), and derives its 2-D outputs from the GATHERED height rows -- so a
wrong row order / wrong window<->storage conversion / missing gather changes the final maps."""
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
        xy = self.xy
        self.height_full = torch.full((xy, 3 * xy), POISON, dtype=torch.float64)
        self.fused_cells = torch.zeros(1, dtype=torch.int64)

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
    def cloud_tensor(self, pc):
        return torch.from_numpy(np.ascontiguousarray(pc[:, :3]))

    def process(self, cloud, ego, tf):
        pc = cloud.numpy()
        if pc.shape[0] == 0:
            self.g.ego_position = ego
            return _gvom.GVOM_EMPTY_CLOUD
        slot = self.g.buffer_index
        was = self.g.origin_buffer[slot]
        with contextlib.redirect_stdout(io.StringIO()):
            self.g.process_pointcloud(pc, ego, tf)
        return _gvom.GVOM_NO_OVERLAP if self.g.origin_buffer[slot] is was else _gvom.GVOM_OK

    def combine_fuse(self):
        with contextlib.redirect_stdout(io.StringIO()):
            out = self.g.combine_maps()
        if out is None:
            return _gvom.GVOM_EMPTY_BUFFER
        g, xy = self.g, self.xy
        self.origin = g.combined_origin
        # positive-obstacle DENSITY (no slope override): the oracle kernel with zero slopes
        zero = np.zeros((xy, xy))
        dens = oracle.make_positive_obstacle_map(g.combined_index_map, g.height_map, xy, self.zs, g.z_resolution,
                                                 g.positive_obstacle_threshold, g.combined_hit_count,
                                                 g.combined_total_count, g.robot_height, self.origin,
                                                 zero, zero, 1e300)
        hf = self.height_full.numpy()
        hf[:] = POISON                                           # other ranks' rows: garbage
        hf[self.lo:self.hi, :xy] = self._to_storage(g.height_map, self.origin)[self.lo:self.hi]
        hf[self.lo:self.hi, xy:2 * xy] = self._to_storage(g.inferred_height_map, self.origin)[self.lo:self.hi]
        hf[self.lo:self.hi, 2 * xy:] = self._to_storage(dens.astype(np.float64), self.origin)[self.lo:self.hi]
        self.fused_cells[0] = self._own_cells(g.combined_index_map, self.origin)
        return _gvom.GVOM_OK

    def set_cell_count(self, n):
        self.cell_count = n

    def combine_map2d(self):
        """ALL rows of the outputs from the gathered [height | inferred | density] rows."""
        g, xy = self.g, self.xy
        hf = self.height_full.numpy()
        h = self._to_window(hf[:, :xy].copy(), self.origin)
        inf = self._to_window(hf[:, xy:2 * xy].copy(), self.origin)
        dens = self._to_window(hf[:, 2 * xy:].copy(), self.origin)
        sx, sy, r = oracle.calculate_slope(h, g.xy_resolution)
        dh = oracle.guess_height(h, inf)
        steep = np.sqrt(sx * sx + sy * sy) >= g.slope_obstacle_threshold
        pos = np.where(steep, 100, dens.astype(np.int32)).astype(np.int32)
        neg = np.where(dh > g.negative_obstacle_threshold, 100, 0).astype(np.int32)
        vis = (h > -1000).astype(np.int32)
        o = self.origin.copy()
        o[0] *= g.xy_resolution; o[1] *= g.xy_resolution; o[2] *= g.z_resolution
        return (o, pos, neg, r, vis)

"""Test doubles for g-vom_amd/gvom_sharded.py so that the ORCHESTRATION of the sharded map (ragged
shares, the host-side count exchange, the sparse all-to-all of accumulator contributions, the global
commit decision, the row all-gather, storage<->window reordering) runs under gloo on CPU:

  * OracleShardBackend: the per-rank compute, with the CPU oracle standing in for the HIP library.
    A rank traces ITS OWN points over the whole window (orc_point_2_map), keeps what falls into its
    own storage rows and ships the rest -- per voxel {index, hit, total, min-height} -- to the owners;
    the owner sums / minimises, encodes its rows and commits.  Every rank only ever PUBLISHES the rows
    of its own slab (everything else is poisoned) and derives its 2-D outputs from the GATHERED rows.
  * GlooComm: the transport, torch.distributed (gloo) in place of RCCL + shared memory.
"""
import contextlib
import io
import math

import numpy as np
import torch
import torch.distributed as dist

import gvom as _gvom
from oracle import oracle

POISON = -7777.0


class GlooComm(object):
    def __init__(self, group=None):
        self.group = group
        self.rank, self.world = dist.get_rank(group), dist.get_world_size(group)

    def exchange_host(self, values):
        mine = torch.tensor([int(v) for v in values], dtype=torch.int64)
        out = [torch.empty_like(mine) for _ in range(self.world)]
        dist.all_gather(out, mine, group=self.group)
        return [[int(x) for x in t] for t in out]

    def barrier(self):
        dist.barrier(group=self.group)

    def exchange_scan(self, backend, send_q, send_e, recv_q, recv_e):
        """send_q[d] / recv_q[s]: "quads" = voxel records of ray passes; send_e / recv_e: endpoints.
        The double ships float64 rows: quads {voxel, total}, endpoints {voxel, hit, min-height}."""
        reqs = []
        inbox_q = [np.zeros((recv_q[s], 2)) for s in range(self.world)]
        inbox_e = [np.zeros((recv_e[s], 3)) for s in range(self.world)]
        for p in range(self.world):
            if p == self.rank:
                continue
            if send_q[p]:
                reqs.append(dist.isend(torch.from_numpy(backend.out_q[p]), p, group=self.group, tag=1))
            if send_e[p]:
                reqs.append(dist.isend(torch.from_numpy(backend.out_e[p]), p, group=self.group, tag=2))
            if recv_q[p]:
                reqs.append(dist.irecv(torch.from_numpy(inbox_q[p]), p, group=self.group, tag=1))
            if recv_e[p]:
                reqs.append(dist.irecv(torch.from_numpy(inbox_e[p]), p, group=self.group, tag=2))
        for r in reqs:
            r.wait()
        backend.in_q, backend.in_e = inbox_q, inbox_e

    def allgather_rows(self, backend):
        full = backend.height_full                      # [xy, 3*xy] tensor, my rows valid
        rows = full.shape[0] // self.world
        mine = full[self.rank * rows:(self.rank + 1) * rows].contiguous()
        out = torch.empty_like(full)
        dist.all_gather_into_tensor(out, mine, group=self.group)
        full.copy_(out)


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
        self.out_q, self.out_e, self.in_q, self.in_e = {}, {}, [], []

    # -- layout helpers ---------------------------------------------------------------------
    def _om(self, origin):
        return int(origin[0]) % self.xy, int(origin[1]) % self.xy

    def _to_storage(self, m_xy, origin):          # reference [x][y] -> storage [sy][sx]
        om0, om1 = self._om(origin)
        return np.roll(np.ascontiguousarray(m_xy.T), (om1, om0), (0, 1))

    def _to_window(self, s_yx, origin):           # storage [sy][sx] -> reference [x][y]
        om0, om1 = self._om(origin)
        return np.ascontiguousarray(np.roll(s_yx, (-om1, -om0), (0, 1)).T)

    def _owner_of_voxels(self, origin):
        """owner rank of every voxel (reference order x + y*xy + z*xy*xy): storage row sy = (y + origin_y) mod xy"""
        xy, zs = self.xy, self.zs
        y = (np.arange(xy * xy * zs) // xy) % xy
        sy = (y + int(origin[1])) % xy
        return sy // self.rows

    # -- interface used by ShardedGvom ----------------------------------------------------------
    def scan_local(self, pointcloud, ego, tf):
        g, L = self.g, oracle.lib()
        g.ego_position = ego
        W = self.world
        pc = oracle._as_cloud(np.asarray(pointcloud)) if len(pointcloud) else np.zeros((0, 3))
        n = pc.shape[0]
        V = g.voxel_count
        origin = np.zeros(3)
        origin[0] = math.floor((ego[0] / g.xy_resolution) - g.xy_size / 2)
        origin[1] = math.floor((ego[1] / g.xy_resolution) - g.xy_size / 2)
        origin[2] = math.floor((ego[2] / g.z_resolution) - g.z_size / 2)
        hit = np.zeros(V, np.int32); total = np.zeros(V, np.int32)
        minh = np.ones(V, np.float32)
        if n:
            suf = "f32" if pc.dtype == np.float32 else "f64"
            if tf is not None:
                t = np.ascontiguousarray(np.asarray(tf, np.float64))
                getattr(L, "orc_transform_pointcloud_" + suf)(oracle._p(pc), n, pc.shape[1], oracle._p(t))
            egoa = np.asarray(ego, dtype=np.float64)
            getattr(L, "orc_point_2_map_" + suf)(g.xy_resolution, g.z_resolution, g.xy_size, g.z_size, g.min_distance,
                                                 oracle._p(pc), n, pc.shape[1], oracle._p(hit), oracle._p(total),
                                                 oracle._p(egoa), oracle._p(origin))
            idx = np.full(V, -1, np.int32)
            cells = L.orc_assign_indices(oracle._p(hit.copy()), oracle._p(total.copy()), oracle._p(idx), V)
            if cells:
                mh = np.ones(cells * 3, np.float32)
                getattr(L, "orc_calculate_min_height_" + suf)(g.xy_resolution, g.z_resolution, g.xy_size, g.z_size,
                                                              g.min_distance, oracle._p(idx), oracle._p(pc), n, pc.shape[1],
                                                              oracle._p(mh), oracle._p(origin))
                occ = idx >= 0
                minh[occ] = mh[idx[occ]]
        owner = self._owner_of_voxels(origin)
        # endpoints: voxels with hit > 0 carry {hit, min-height} and their endpoint share of total (= hit);
        # "quads": the ray passes, total - hit
        passes = total - hit
        send_q, send_e = [0] * W, [0] * W
        self.out_q, self.out_e = {}, {}
        for d in range(W):
            if d == self.rank:
                continue
            vq = np.nonzero((owner == d) & (passes > 0))[0]
            ve = np.nonzero((owner == d) & (hit > 0))[0]
            self.out_q[d] = np.stack([vq.astype(np.float64), passes[vq].astype(np.float64)], 1) if len(vq) else np.zeros((0, 2))
            self.out_e[d] = np.stack([ve.astype(np.float64), hit[ve].astype(np.float64), minh[ve].astype(np.float64)], 1) if len(ve) else np.zeros((0, 3))
            send_q[d], send_e[d] = len(vq), len(ve)
        mine = owner == self.rank
        self._acc = (np.where(mine, hit, 0).astype(np.int32), np.where(mine, total, 0).astype(np.int32),
                     np.where(mine, minh, np.float32(1.0)).astype(np.float32), origin)
        return send_q, send_e, int(hit.any()), n

    def recv_reserve(self, recv_eps):
        pass

    def scan_merge(self, recv_quads, recv_eps, accept):
        g, L = self.g, oracle.lib()
        hit, total, minh, origin = self._acc
        for s in range(self.world):
            if s == self.rank:
                continue
            if recv_quads[s]:
                q = self.in_q[s]
                np.add.at(total, q[:, 0].astype(np.int64), q[:, 1].astype(np.int32))
            if recv_eps[s]:
                e = self.in_e[s]
                v = e[:, 0].astype(np.int64)
                np.add.at(hit, v, e[:, 1].astype(np.int32))
                np.add.at(total, v, e[:, 1].astype(np.int32))
                np.minimum.at(minh, v, e[:, 2].astype(np.float32))
        if not accept:
            return
        V = g.voxel_count
        index_map = np.full(V, -1, np.int32)
        cells = L.orc_assign_indices(oracle._p(hit), oracle._p(total), oracle._p(index_map), V)
        chit = np.empty(cells, np.int32); ctotal = np.empty(cells, np.int32)
        L.orc_move_data(oracle._p(hit), oracle._p(chit), oracle._p(index_map), V)
        L.orc_move_data(oracle._p(total), oracle._p(ctotal), oracle._p(index_map), V)
        min_height = np.ones(max(cells, 1) * 3, np.float32)
        occ = index_map >= 0
        min_height[index_map[occ]] = minh[occ]
        b = g.buffer_index                                                                   # gvom.py:163-175
        g.metrics_buffer[b] = None
        g.index_buffer[b] = index_map
        g.hit_count_buffer[b] = chit
        g.total_count_buffer[b] = ctotal
        g.min_height_buffer[b] = min_height
        g.origin_buffer[b] = origin
        g.last_buffer_index = b
        g.buffer_index = (b + 1) % g.buffer_size

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
        return _gvom.GVOM_OK

    def local_fused_cells(self):
        return int(self.g.combined_cell_count_cpu)

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

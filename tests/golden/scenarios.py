"""Scenario definitions + a generic runner/recorder for golden vectors.

Used by tests/golden/make_golden.py (conda python3.9: drives the REFERENCE under the Numba
CUDA simulator and records its outputs) and by the tests (drive the oracle / the HIP
library over the recorded inputs and compare).  Works with any object exposing the
reference's `Gvom` surface; device arrays are read through `.copy_to_host()` when present.

A scenario is a dict: {"params": (14 ctor args), "steps": [("scan", pc, ego, tf|None) |
("combine",)]}.  The recorder flattens everything into one .npz (inputs AND outputs), so
tests never depend on RNG reproducibility across numpy versions.
"""
import numpy as np

REF_DEFAULT_TAIL = (1.0, 0.5, 0.5, 0.3, 2.0, 4.0, 1.0, 1, 1)   # gvom_ros.py:29-41 defaults


def host(a):
    if a is None:
        return None
    if hasattr(a, "copy_to_host"):
        return a.copy_to_host()
    return np.asarray(a)


def dense_from_compact(index_map, hit, total, min_height):
    index_map = np.asarray(index_map)
    occ = index_map >= 0
    state = np.where(occ, 0, index_map).astype(np.int32)
    hd = np.zeros(index_map.shape, np.int32); td = np.zeros(index_map.shape, np.int32)
    md = np.ones(index_map.shape, np.float32)
    rows = index_map[occ]
    hd[occ] = np.asarray(hit)[rows]; td[occ] = np.asarray(total)[rows]
    md[occ] = np.asarray(min_height)[rows]
    return state, hd, td, md


# --------------------------------------------------------------------------------------
# synthetic clouds
# --------------------------------------------------------------------------------------

def uniform_cloud(rng, n, xlim, ylim, zlim, dtype=np.float64):
    x = rng.uniform(xlim[0], xlim[1], n)
    y = rng.uniform(ylim[0], ylim[1], n)
    z = rng.uniform(zlim[0], zlim[1], n)
    return np.stack([x, y, z], axis=1).astype(dtype)


def lidar_on_terrain(sensor, n_az, n_el, el_lo, el_hi, terrain, max_range, noise=0.0, rng=None,
                     dtype=np.float64, walls=()):
    """Casts n_el x n_az rays from `sensor` onto a height field terrain(x, y)->z by marching
    (coarse, deterministic); optional axis-aligned box walls ((x0,x1,y0,y1,z0,z1), ...).
    Returns world-frame hit points (rays without a return are dropped)."""
    pts = []
    for ie in range(n_el):
        el = el_lo + (el_hi - el_lo) * ie / max(n_el - 1, 1)
        for ia in range(n_az):
            az = 2.0 * np.pi * ia / n_az
            d = np.array([np.cos(el) * np.cos(az), np.cos(el) * np.sin(az), np.sin(el)])
            t = 0.05
            hit = None
            while t < max_range:
                p = sensor + t * d
                inside_wall = False
                for (x0, x1, y0, y1, z0, z1) in walls:
                    if x0 <= p[0] <= x1 and y0 <= p[1] <= y1 and z0 <= p[2] <= z1:
                        inside_wall = True
                if p[2] <= terrain(p[0], p[1]) or inside_wall:
                    hit = p
                    break
                t += 0.02
            if hit is not None:
                if noise > 0.0 and rng is not None:
                    hit = hit + rng.normal(0.0, noise, 3)
                pts.append(hit)
    return np.asarray(pts, dtype=dtype).reshape(-1, 3)


def rot_z(angle, t):
    c, s = np.cos(angle), np.sin(angle)
    T = np.eye(4)
    T[0, 0] = c; T[0, 1] = -s; T[1, 0] = s; T[1, 1] = c
    T[:3, 3] = t
    return T


# --------------------------------------------------------------------------------------
# scenarios F1..F7
# --------------------------------------------------------------------------------------

def scenario_f1():
    rng = np.random.default_rng(0)
    pc = uniform_cloud(rng, 300, (-3, 3), (-3, 3), (-1.5, 0.5))
    return {"params": (0.4, 0.4, 16, 8, 2) + REF_DEFAULT_TAIL,
            "steps": [("scan", pc, (0.0, 0.0, 0.0), None), ("combine",)]}


def scenario_f2():
    rng = np.random.default_rng(1)
    pc = uniform_cloud(rng, 300, (-3, 3), (-3, 3), (-1.2, 0.8), dtype=np.float32)
    # general rigid transform: yaw + small roll, translation
    T = rot_z(0.3, (0.37, -0.21, 0.11))
    roll = np.eye(4); a = 0.05
    roll[1, 1] = np.cos(a); roll[1, 2] = -np.sin(a); roll[2, 1] = np.sin(a); roll[2, 2] = np.cos(a)
    T = T @ roll
    return {"params": (0.4, 0.2, 16, 8, 1) + REF_DEFAULT_TAIL,
            "steps": [("scan", pc, (0.37, -0.21, 0.11), T), ("combine",)]}


def scenario_f3():
    """3 scans, moving ego (origin shifts in x, y and z), buffer_size=2, combine after each,
    plus a 4th combine with no new scan (count compounding, ring wrap, decay rule)."""
    rng = np.random.default_rng(3)
    params = (0.4, 0.2, 20, 12, 2, 0.5, 0.5, 0.5, 0.3, 2.0, 2.0, 1.0, 1, 1)
    egos = [(0.0, 0.0, 0.0), (0.9, -0.5, 0.25), (1.7, 0.45, -0.3)]
    steps = []
    for k, ego in enumerate(egos):
        base = uniform_cloud(rng, 500, (-3.5, 3.5), (-3.5, 3.5), (-1.0, 0.2))
        # a persistent dense blob (so hit>10 voxels and decay both show up) + per-scan noise
        blob = uniform_cloud(rng, 150, (1.2, 1.6), (0.8, 1.2), (-0.6, 0.4))
        pc = np.concatenate([base + np.array(ego), blob], axis=0)
        if k == 2:
            pc = pc[:500]      # third scan loses the blob: free-space rays now cross it
        steps.append(("scan", pc, ego, None))
        steps.append(("combine",))
    steps.append(("combine",))
    return {"params": params, "steps": steps}


def scenario_f4():
    """Drop-off: flat ground then a cliff -> shadow zone with observed-free voxels but no
    ground returns (inferred height, __guess_height typos, negative obstacles)."""
    def terrain(x, y):
        return -1.0 if x < 1.5 else -2.6
    sensor = np.array([0.1, 0.05, 0.0])
    pc = lidar_on_terrain(sensor, 96, 14, np.deg2rad(-60), np.deg2rad(-6), terrain, 9.0)
    params = (0.4, 0.2, 24, 16, 1, 0.5, 0.5, 0.5, 0.3, 2.0, 1.0, 1.0, 1, 1)
    return {"params": params, "steps": [("scan", pc, tuple(sensor), None), ("combine",)]}


def scenario_f5():
    """Steep ramp + a wall with many returns per voxel (slope>=thr -> 100; hit>10 density)."""
    def terrain(x, y):
        return -1.0 + max(0.0, (y - 0.8)) * 0.9       # ramp of slope 0.9 for y > 0.8
    sensor = np.array([0.0, 0.0, 0.0])
    walls = ((1.8, 2.2, -1.5, 0.5, -1.0, 1.0),)
    rng = np.random.default_rng(5)
    pc = lidar_on_terrain(sensor, 160, 24, np.deg2rad(-55), np.deg2rad(20), terrain, 6.0,
                          noise=0.01, rng=rng, walls=walls)
    params = (0.4, 0.2, 16, 16, 1, 0.5, 0.5, 0.5, 0.3, 2.0, 1.0, 1.0, 1, 1)
    return {"params": params, "steps": [("scan", pc, tuple(sensor), None), ("combine",)]}


def scenario_f6():
    """Degenerate inputs: empty ring combine, empty cloud, all out of grid, all inside
    min_distance, point == ego, then one valid scan."""
    rng = np.random.default_rng(6)
    params = (0.4, 0.4, 16, 8, 2) + REF_DEFAULT_TAIL
    far = uniform_cloud(rng, 50, (50, 60), (50, 60), (20, 30))
    near = uniform_cloud(rng, 50, (-0.3, 0.3), (-0.3, 0.3), (-0.3, 0.3))
    ego = (1.3, 0.2, 0.1)
    same = np.array([[1.3, 0.2, 0.1], [1.3, 0.2, 0.1]])
    ok = uniform_cloud(rng, 200, (-2, 3), (-2, 3), (-1.0, 0.5))
    return {"params": params,
            "steps": [("combine",),
                      ("scan", np.zeros((0, 3)), ego, None),
                      ("scan", far, ego, None),
                      ("scan", near, (0.0, 0.0, 0.0), None),
                      ("combine",),
                      ("scan", same, ego, None),
                      ("combine",),
                      ("scan", ok, ego, None),
                      ("combine",)]}


def scenario_f7():
    """BASELINE c1: 64x64x32, 50,000 uniform points, seed 1234."""
    rng = np.random.default_rng(1234)
    n = 50000
    x = rng.uniform(-14, 14, n); y = rng.uniform(-14, 14, n); z = rng.uniform(-3.5, 3.5, n)
    pc = np.stack([x, y, z], axis=1)
    params = (0.4, 0.2, 64, 32, 1) + REF_DEFAULT_TAIL
    return {"params": params, "steps": [("scan", pc, (0.3, -0.2, 0.1), None), ("combine",)]}


SCENARIOS = {"f1": scenario_f1, "f2": scenario_f2, "f3": scenario_f3, "f4": scenario_f4,
             "f5": scenario_f5, "f6": scenario_f6, "f7": scenario_f7}

MAPS_2D = ("height_map", "inferred_height_map", "x_slope_map", "y_slope_map", "roughness_map",
           "guessed_height_delta")


# --------------------------------------------------------------------------------------
# runner / recorder
# --------------------------------------------------------------------------------------

def run_and_record(make_gvom, scenario, record_debug=True):
    """Drives `make_gvom(*params)` through the scenario; returns a flat dict of arrays."""
    rec = {"params": np.asarray(scenario["params"], dtype=np.float64),
           "n_steps": np.asarray(len(scenario["steps"]))}
    g = make_gvom(*scenario["params"])
    for k, step in enumerate(scenario["steps"]):
        pre = "s%d_" % k
        if step[0] == "scan":
            _, pc, ego, tf = step
            rec[pre + "kind"] = np.asarray(0)
            rec[pre + "pc"] = np.asarray(pc)
            rec[pre + "ego"] = np.asarray(ego, dtype=np.float64)
            if tf is not None:
                rec[pre + "tf"] = np.asarray(tf, dtype=np.float64)
            pc_in = np.array(pc, copy=True)
            g.process_pointcloud(pc_in, tuple(float(e) for e in ego),
                                 None if tf is None else np.array(tf, copy=True))
            assert np.array_equal(pc_in, np.asarray(pc)), "input cloud was mutated"
            rec[pre + "buffer_index"] = np.asarray(g.buffer_index)
            rec[pre + "last_buffer_index"] = np.asarray(g.last_buffer_index)
            b = g.last_buffer_index
            filled = np.asarray([o is not None for o in g.origin_buffer])
            rec[pre + "slots_filled"] = filled
            if g.origin_buffer[b] is not None:
                st, hd, td, md = dense_from_compact(host(g.index_buffer[b]),
                                                    host(g.hit_count_buffer[b]),
                                                    host(g.total_count_buffer[b]),
                                                    host(g.min_height_buffer[b]))
                rec[pre + "slot"] = np.asarray(b)
                rec[pre + "slot_state"] = st
                rec[pre + "slot_hit"] = hd
                rec[pre + "slot_total"] = td
                rec[pre + "slot_min_h"] = md
                rec[pre + "slot_origin"] = host(g.origin_buffer[b]).astype(np.float64)
        else:
            rec[pre + "kind"] = np.asarray(1)
            out = g.combine_maps()
            rec[pre + "returned_none"] = np.asarray(out is None)
            if out is None:
                continue
            rec[pre + "origin_world"] = np.asarray(out[0])
            rec[pre + "positive"] = np.asarray(out[1])
            rec[pre + "negative"] = np.asarray(out[2])
            rec[pre + "roughness"] = np.asarray(out[3])
            rec[pre + "visibility"] = np.asarray(out[4])
            st, hd, td, md = dense_from_compact(host(g.combined_index_map),
                                                host(g.combined_hit_count),
                                                host(g.combined_total_count),
                                                host(g.combined_min_height))
            rec[pre + "fused_state"] = st
            rec[pre + "fused_hit"] = hd
            rec[pre + "fused_total"] = td
            rec[pre + "fused_min_h"] = md
            rec[pre + "cell_count"] = np.asarray(int(g.combined_cell_count_cpu))
            for name in MAPS_2D:
                rec[pre + name] = host(getattr(g, name)).astype(np.float64)
            if record_debug:
                rec[pre + "occupancy"] = np.asarray(g.get_map_as_occupancy_grid())
                rec[pre + "debug_height_map"] = np.asarray(g.make_debug_height_map())
                rec[pre + "debug_inferred_height_map"] = np.asarray(g.make_debug_inferred_height_map())
                vox = g.make_debug_voxel_map()
                if vox is not None:                  # rows = compact index (unspecified order): sort by xyz
                    vox = np.asarray(vox)
                    order = np.lexsort((vox[:, 2], vox[:, 1], vox[:, 0]))
                    rec[pre + "debug_voxel_map"] = vox[order]
    return rec


def scenario_from_record(rec):
    """Rebuilds {"params", "steps"} from a recorded .npz (inputs only)."""
    p = rec["params"]
    params = tuple(float(v) for v in p[:2]) + tuple(int(v) for v in p[2:5]) + \
        tuple(float(v) for v in p[5:12]) + tuple(int(v) for v in p[12:14])
    steps = []
    for k in range(int(rec["n_steps"])):
        pre = "s%d_" % k
        if int(rec[pre + "kind"]) == 0:
            tf = rec[pre + "tf"] if (pre + "tf") in rec else None
            steps.append(("scan", rec[pre + "pc"], tuple(rec[pre + "ego"].tolist()), tf))
        else:
            steps.append(("combine",))
    return {"params": params, "steps": steps}

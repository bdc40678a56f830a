"""Seeded synthetic inputs for the BASELINE.json configurations.

Used by bench.py and the tests; there is no network for datasets.  A rotating multi-beam
lidar (Ouster OS1-64 / OS1-128 shaped: `beams` elevations x 2048 azimuths, beam-major
order, elevation uniform in +-22.5 deg) is ray-cast analytically against a scene made of a
noisy ground plane at z = -1 m, one 1 m deep trench and 20 seeded boxes.  Rays without a
return within 60 m are CLAMPED to 60 m (not dropped) so that every scan has exactly
beams*2048 points, as the configs are quoted.
"""
import numpy as np

MAX_RANGE = 60.0


def make_scene(seed=2, n_boxes=20, extent=22.0):
    rng = np.random.default_rng(seed)
    boxes = []
    while len(boxes) < n_boxes:
        cx, cy = rng.uniform(-extent, extent, 2)
        if np.hypot(cx, cy) < 2.5:
            continue                      # keep the robot's own footprint free
        sx, sy, sz = rng.uniform(0.5, 3.0, 3)
        boxes.append((cx - sx / 2, cx + sx / 2, cy - sy / 2, cy + sy / 2, -1.0, -1.0 + sz))
    trench = (6.0, 8.0, -12.0, 12.0, -2.0)          # x0, x1, y0, y1, floor z
    return {"boxes": np.asarray(boxes, np.float64), "trench": trench, "ground_z": -1.0, "seed": seed}


def _ray_box(o, d, box):
    """Slab test, vectorised over rays; returns entry distance (inf if missed)."""
    with np.errstate(divide="ignore", invalid="ignore"):
        inv = 1.0 / d
        t0 = (np.array([box[0], box[2], box[4]]) - o) * inv
        t1 = (np.array([box[1], box[3], box[5]]) - o) * inv
    tmin = np.nanmax(np.minimum(t0, t1), axis=1)
    tmax = np.nanmin(np.maximum(t0, t1), axis=1)
    hit = (tmax >= np.maximum(tmin, 0.0)) & (tmin > 1e-6)
    return np.where(hit, tmin, np.inf)


def os1_like_elevations(beams, seed=7):
    """Beam elevations (degrees) of a REAL multi-beam lidar rather than a uniform comb: denser around the horizon (a
    "gradient" beam configuration), every beam off its nominal angle by a fixed calibration error of up to +-0.25 deg --
    no two neighbouring beams are the same distance apart."""
    u = np.linspace(-1.0, 1.0, beams)
    el = 22.5 * np.sign(u) * np.abs(u) ** 1.6
    el += np.random.default_rng(seed).uniform(-0.25, 0.25, beams)
    return np.sort(el)


def os1_like_azimuth_offsets(beams):
    """Per-beam azimuth offsets (radians): the beams of such a sensor sit in four staggered columns, ~ +-3 deg apart."""
    return np.deg2rad(np.array([-3.1, -1.0, 1.0, 3.1])[np.arange(beams) % 4])


def drop_returns(cloud, fraction, seed):
    """What a real node hands over (gvom_ros.py:108: ros_numpy's xyz array has the invalid returns REMOVED): `fraction` of the
    returns taken out before the call -- half of them in bursts of 8..64 consecutive returns of a beam (glass, absorbers, the
    robot's own body), half singly -- the order of the survivors kept.  The length differs from scan to scan (seed)."""
    n = cloud.shape[0]
    rng = np.random.default_rng(5000 + seed)
    keep = np.ones(n, bool)
    target = int(fraction * n)
    removed = 0
    while removed < target // 2:
        a = int(rng.integers(0, n))
        ln = int(rng.integers(8, 65))
        seg = keep[a:a + ln]
        removed += int(seg.sum())
        seg[:] = False
    rest = target - int((~keep).sum())
    if rest > 0:
        alive = np.flatnonzero(keep)
        keep[rng.choice(alive, size=min(rest, alive.size), replace=False)] = False
    return np.ascontiguousarray(cloud[keep])


def lidar_scan(scene, beams=64, azimuths=2048, sensor=(0.0, 0.0, 0.0), yaw=0.0, noise_seed=0,
               dtype=np.float32, frame="world", elevations_deg=None, azimuth_offsets=None):
    """Returns an (beams*azimuths, 3) cloud, beam-major.  frame="world": points in the world
    frame (transform=None); frame="sensor": points in the sensor frame, use with
    `sensor_transform(sensor, yaw)`.  elevations_deg / azimuth_offsets: per-beam elevations and azimuth offsets of a real
    sensor (os1_like_*) instead of the uniform comb."""
    o = np.asarray(sensor, np.float64)
    el = np.deg2rad(np.linspace(-22.5, 22.5, beams) if elevations_deg is None else np.asarray(elevations_deg, np.float64))
    az = 2.0 * np.pi * np.arange(azimuths) / azimuths + yaw
    az2 = az[None, :] + (0.0 if azimuth_offsets is None else np.asarray(azimuth_offsets, np.float64)[:, None])
    ce, se = np.cos(el)[:, None], np.sin(el)[:, None]
    d = np.stack([ce * np.cos(az2), ce * np.sin(az2), se * np.ones_like(az2)],
                 axis=-1).reshape(-1, 3)
    n = d.shape[0]
    rng = np.random.default_rng(1000 + noise_seed)
    t = np.full(n, np.inf)
    # ground plane (with 2 cm noise), opened up over the trench
    gz = scene["ground_z"] + rng.normal(0.0, 0.02, n)
    with np.errstate(divide="ignore", invalid="ignore"):
        tg = (gz - o[2]) / d[:, 2]
    tg = np.where((d[:, 2] < 0) & (tg > 0), tg, np.inf)
    with np.errstate(invalid="ignore"):
        pg = o + tg[:, None] * d
    x0, x1, y0, y1, fz = scene["trench"]
    in_trench = (pg[:, 0] > x0) & (pg[:, 0] < x1) & (pg[:, 1] > y0) & (pg[:, 1] < y1)
    t = np.where(in_trench, np.inf, tg)
    # trench floor and walls as an inverted box: hit the floor or the far walls
    with np.errstate(divide="ignore", invalid="ignore"):
        tf_ = (fz - o[2]) / d[:, 2]
    with np.errstate(invalid="ignore"):
        pf = o + tf_[:, None] * d
    floor_hit = in_trench & (d[:, 2] < 0) & (pf[:, 0] > x0) & (pf[:, 0] < x1) & (pf[:, 1] > y0) & (pf[:, 1] < y1)
    t = np.where(floor_hit, tf_, t)
    # rays that enter the trench mouth but leave the floor footprint hit a wall: approximate by
    # the exit of the trench volume
    wall = in_trench & ~floor_hit
    if np.any(wall):
        box = (x0, x1, y0, y1, fz, scene["ground_z"])
        with np.errstate(divide="ignore", invalid="ignore"):
            inv = 1.0 / d[wall]
            ta = (np.array([box[0], box[2], box[4]]) - o) * inv
            tb = (np.array([box[1], box[3], box[5]]) - o) * inv
        texit = np.nanmin(np.maximum(ta, tb), axis=1)
        tw = t.copy(); tw[wall] = texit
        t = tw
    for b in scene["boxes"]:
        t = np.minimum(t, _ray_box(o, d, b))
    t = np.where(np.isfinite(t) & (t < MAX_RANGE), t, MAX_RANGE)
    pts = o + t[:, None] * d
    if frame == "sensor":
        T = sensor_transform(sensor, yaw=0.0)
        Ti = np.linalg.inv(T)
        pts = pts @ Ti[:3, :3].T + Ti[:3, 3]
    return np.ascontiguousarray(pts.astype(dtype))


def sensor_transform(sensor, yaw=0.0):
    c, s = np.cos(yaw), np.sin(yaw)
    T = np.eye(4)
    T[0, 0] = c; T[0, 1] = -s; T[1, 0] = s; T[1, 1] = c
    T[:3, 3] = sensor
    return T


def uniform_cloud(n, seed, xlim, ylim, zlim, dtype=np.float64):
    """BASELINE c1 style cloud: x, then y, then z drawn from one default_rng(seed)."""
    rng = np.random.default_rng(seed)
    x = rng.uniform(xlim[0], xlim[1], n)
    y = rng.uniform(ylim[0], ylim[1], n)
    z = rng.uniform(zlim[0], zlim[1], n)
    return np.stack([x, y, z], axis=1).astype(dtype)


REF_TAIL = (1.0, 0.5, 0.5, 0.3, 2.0, 4.0, 1.0, 1, 1)        # reference defaults, gvom_ros.py:29-41

CONFIGS = {
    # name: (ctor params, beams, description)
    "c1": ((0.4, 0.2, 64, 32, 1) + REF_TAIL, None, "64x64x32, 50k uniform points"),
    "c2": ((0.2, 0.2, 256, 64, 1) + REF_TAIL, 64, "256x256x64 @0.2 m, OS1-64 131,072-pt scan, buffer=1"),
    "c3": ((0.2, 0.2, 256, 64, 8) + REF_TAIL, 128, "256x256x64 @0.2 m, OS1-128 262,144-pt scan, buffer=8"),
    "m256": ((0.2, 0.2, 256, 256, 1) + REF_TAIL, 64, "256x256x256 @0.2 m (metric grid), OS1-64 131,072-pt scan, buffer=1"),
    "m256b8": ((0.2, 0.2, 256, 256, 8) + REF_TAIL, 64, "256x256x256 @0.2 m (metric grid), OS1-64 scan, buffer=8"),
    # the multi-GPU configs of BASELINE.json, also runnable on ONE GPU (sensors = 128-beam scans interleaved in azimuth)
    # (buffer sizes as BASELINE.md section 3 states them: c4 buffer=4, c5 buffer=8)
    "c4": ((0.2, 0.2, 512, 128, 4) + REF_TAIL, 128, "512x512x128 @0.2 m, 4 x OS1-128 = 1,048,576-pt cloud, buffer=4"),
    "c5": ((0.2, 0.2, 1024, 128, 8) + REF_TAIL, 128, "1024x1024x128 @0.2 m, 16 x OS1-128 = 4,194,304-pt cloud per tick, buffer=8"),
}
# what a real node delivers: the metric grid, an OS1-64-like sensor with NON-UNIFORM beam elevations and
# staggered beam columns, 10 / 25 / 40 % of the returns removed before the call, a different length every scan
for _pct in (10, 25, 40):
    CONFIGS["m256_d%d" % _pct] = (CONFIGS["m256"][0], 64, "256x256x256 @0.2 m (metric grid), OS1-64-like scan with non-uniform beam "
                                  "elevations, %d %% of the 131,072 returns dropped before the call (length varies per scan), buffer=1" % _pct)
SENSORS = {"c4": 4, "c5": 16}


def config_inputs(name, n_scans=1, dtype=np.float32):
    """(params, [(cloud, ego, transform), ...]) for a named BASELINE config; successive scans
    move the sensor 0.2 m in +x."""
    params, beams, _ = CONFIGS[name]
    scans = []
    if name == "c1":
        pc = uniform_cloud(50000, 1234, (-14, 14), (-14, 14), (-3.5, 3.5))
        return params, [(pc, (0.3, -0.2, 0.1), None)]
    if name in SENSORS:
        nsens = SENSORS[name]
        scene = make_scene(2, extent=0.2 * params[2] / 2 * 0.9)
        for k in range(n_scans):
            sensor = (0.2 * k, -0.1 * k, 0.0)
            scans.append((np.concatenate([lidar_scan(scene, beams=beams, sensor=sensor, yaw=2 * np.pi / 2048 * r / nsens,
                                                     noise_seed=100 * k + r, dtype=dtype) for r in range(nsens)], 0),
                          sensor, None))
        return params, scans
    scene = make_scene(2)
    if name.startswith("m256_d"):
        frac = int(name[6:]) / 100.0
        el, azo = os1_like_elevations(beams), os1_like_azimuth_offsets(beams)
        for k in range(n_scans):
            sensor = (0.2 * k, 0.0, 0.0)
            full = lidar_scan(scene, beams=beams, sensor=sensor, noise_seed=k, dtype=dtype, elevations_deg=el, azimuth_offsets=azo)
            # (the fraction itself wanders by a tenth from scan to scan, as a scene's share of invalid returns does)
            scans.append((drop_returns(full, frac * (0.9 + 0.2 * ((k * 37) % 11) / 10.0), k), sensor, None))
        return params, scans
    for k in range(n_scans):
        sensor = (0.2 * k, 0.0, 0.0)
        scans.append((lidar_scan(scene, beams=beams, sensor=sensor, noise_seed=k, dtype=dtype),
                      sensor, None))
    return params, scans

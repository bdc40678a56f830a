#!/opt/conda/bin/python3.9
"""Generates the golden vectors under tests/golden/ by running the UNMODIFIED reference
(/root/reference/scripts/gvom.py) under Numba's CUDA simulator.

THIS CONTAINER ONLY:   /opt/conda/bin/python3.9 tests/golden/make_golden.py [names...]
  names: f1 f2 f3 f4 f5 f6 f7 kat   (default: everything except f7, which takes ~1 h)

Only data (inputs + the reference's outputs) is written; no reference code is copied.
"""
import os
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_shim                              # noqa: E402
gvom = ref_shim.load_reference()             # the reference module
import numpy as np                           # noqa: E402
from numba import cuda                       # noqa: E402
import scenarios                             # noqa: E402

K = gvom.Gvom                                # kernels are name-mangled staticmethods


def kern(name):
    return getattr(K, "_Gvom__" + name)


def save(name, rec):
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **rec)
    print("wrote %s (%d arrays, %.1f KiB)" % (path, len(rec), os.path.getsize(path) / 1024.0))


# ---------------------------------------------------------------------------------------
# kernel-level known-answer vectors
# ---------------------------------------------------------------------------------------

def ref_point_2_map(xy_res, z_res, xy, zs, min_distance, pts, ego, origin):
    V = xy * xy * zs
    hit = cuda.to_device(np.zeros(V, np.int32))
    total = cuda.to_device(np.zeros(V, np.int32))
    n = pts.shape[0]
    tpb = 32
    kern("point_2_map")[int(np.ceil(n / tpb)), tpb](
        xy_res, z_res, xy, zs, min_distance, cuda.to_device(pts), hit, total, n,
        cuda.to_device(np.asarray(ego, dtype=np.float64)),
        cuda.to_device(np.asarray(origin, dtype=np.float64)))
    return hit.copy_to_host(), total.copy_to_host()


def kat_point_2_map():
    cases = []
    e = (0.5, 0.5, 0.5)
    o = (0.0, 0.0, 0.0)

    def case(name, ego, pt, res=(1.0, 1.0), origin=o, md=0.0, dtype=np.float64, size=(8, 4)):
        cases.append(dict(name=name, xy_res=res[0], z_res=res[1], xy=size[0], zs=size[1], md=md,
                          pts=np.asarray([pt], dtype=dtype), ego=ego, origin=origin))
    case("A", e, (5.5, 0.5, 0.5))
    case("B", e, (5.9, 0.5, 0.5))
    case("C", e, (4.5, 4.5, 0.5))
    case("C32", e, (4.5, 4.5, 0.5), dtype=np.float32)
    case("D", e, (20.5, 0.5, 0.5))
    case("E", e, e)
    case("F", e, (1.2, 0.5, 0.5))
    case("G", (0.0, 0.0, 0.0), (2.0, 0.0, 0.0), md=3.0)
    case("H", (2.5, 2.5, 3.5), (2.6, 2.7, 0.5))
    case("J", e, (3.5, 3.5, 3.5))
    case("K", (0.1, 0.1, 0.1), (1.0, 0.3, -0.2), res=(0.4, 0.2), origin=(-4.0, -4.0, -2.0), md=0.5)
    # negative directions, ties between x and y, tiny rays, off-grid ego
    case("L", (6.5, 6.5, 2.5), (0.5, 1.5, 0.5))
    case("M", (3.5, 3.5, 1.5), (0.5, 6.5, 1.5))
    case("N", (3.5, 3.5, 1.5), (3.5000001, 3.5, 1.5))
    case("O", (-3.5, 3.5, 1.5), (6.5, 3.2, 1.9))
    rec = {"n_cases": np.asarray(len(cases))}
    for k, c in enumerate(cases):
        hit, total = ref_point_2_map(c["xy_res"], c["z_res"], c["xy"], c["zs"], c["md"], c["pts"],
                                     c["ego"], c["origin"])
        pre = "c%d_" % k
        rec[pre + "name"] = np.asarray(c["name"])
        rec[pre + "scal"] = np.asarray([c["xy_res"], c["z_res"], c["xy"], c["zs"], c["md"]], np.float64)
        rec[pre + "pts"] = c["pts"]
        rec[pre + "ego"] = np.asarray(c["ego"], np.float64)
        rec[pre + "origin"] = np.asarray(c["origin"], np.float64)
        rec[pre + "hit"] = hit
        rec[pre + "total"] = total
    # random bundles: awkward resolutions, f32 and f64 clouds, ego off voxel centres
    rng = np.random.default_rng(42)
    k = len(cases)
    for dtype in (np.float64, np.float32):
        for (xy_res, z_res, xy, zs) in ((0.4, 0.2, 12, 6), (0.2, 0.2, 16, 8), (0.37, 0.13, 10, 7)):
            ego = rng.uniform(-0.5, 0.5, 3)
            origin = np.floor(np.array([ego[0] / xy_res - xy / 2, ego[1] / xy_res - xy / 2,
                                        ego[2] / z_res - zs / 2]))
            half = xy * xy_res * 0.7
            pts = np.stack([rng.uniform(-half, half, 96), rng.uniform(-half, half, 96),
                            rng.uniform(-zs * z_res * 0.7, zs * z_res * 0.7, 96)], axis=1).astype(dtype)
            hit, total = ref_point_2_map(xy_res, z_res, xy, zs, 0.3, pts, ego, origin)
            pre = "c%d_" % k
            rec[pre + "name"] = np.asarray("rand_%s_%g" % (np.dtype(dtype).name, xy_res))
            rec[pre + "scal"] = np.asarray([xy_res, z_res, xy, zs, 0.3], np.float64)
            rec[pre + "pts"] = pts
            rec[pre + "ego"] = ego
            rec[pre + "origin"] = origin
            rec[pre + "hit"] = hit
            rec[pre + "total"] = total
            k += 1
    rec["n_cases"] = np.asarray(k)
    save("kat_point_2_map", rec)


def kat_transform():
    rng = np.random.default_rng(7)
    rec = {}
    T = scenarios.rot_z(1.1, (3.3, -7.7, 0.9))
    T[2, 0] = 0.01; T[0, 2] = -0.013
    for name, dtype in (("f32", np.float32), ("f64", np.float64)):
        pts = (rng.normal(0, 10, (64, 3))).astype(dtype)
        d = cuda.to_device(pts)
        kern("transform_pointcloud")[1, 64](d, T, 64)
        rec[name + "_in"] = pts
        rec[name + "_out"] = d.copy_to_host()
    rec["T"] = T
    save("kat_transform", rec)


def _blocks2d(xy):
    return (int(np.ceil(xy / 16)), int(np.ceil(xy / 16)))


def ref_slope(h, xy_res):
    xy = h.shape[0]
    sx = cuda.to_device(np.zeros((xy, xy))); sy = cuda.to_device(np.zeros((xy, xy)))
    r = cuda.to_device(np.full((xy, xy), -1.0))
    kern("calculate_slope")[_blocks2d(xy), (16, 16)](cuda.to_device(h), xy, xy_res, sx, sy, r)
    return sx.copy_to_host(), sy.copy_to_host(), r.copy_to_host()


def ref_guess(h, inf):
    xy = h.shape[0]
    out = cuda.to_device(np.zeros((xy, xy)))
    z = cuda.to_device(np.zeros((xy, xy)))
    kern("guess_height")[_blocks2d(xy), (16, 16)](cuda.to_device(h), cuda.to_device(inf), xy, 0.4, z, z, out)
    return out.copy_to_host()


def kat_2d():
    rec = {}
    xy = 8
    # S1 exact plane, S2 three flat cells, S3 collinear, S4 random rough patch
    h1 = np.fromfunction(lambda x, y: 0.25 * (0.4 * x) - 0.1 * (0.4 * y), (xy, xy))
    h2 = np.full((xy, xy), -1000.0); h2[3, 3] = h2[3, 4] = h2[4, 3] = 0.0
    h3 = np.full((xy, xy), -1000.0); h3[3, 2] = 0.1; h3[3, 3] = 0.2; h3[3, 4] = 0.3
    rng = np.random.default_rng(11)
    h4 = rng.normal(0, 0.3, (xy, xy)); h4[rng.uniform(size=(xy, xy)) < 0.35] = -1000.0
    for name, h in (("S1", h1), ("S2", h2), ("S3", h3), ("S4", h4)):
        sx, sy, r = ref_slope(h, 0.4)
        rec[name + "_h"] = h; rec[name + "_sx"] = sx; rec[name + "_sy"] = sy; rec[name + "_r"] = r
    # G1..G5 + G6 random
    def gcase(name, setters, inf_val=0.5, own=None):
        h = np.full((xy, xy), -1000.0); inf = np.full((xy, xy), -1000.0)
        inf[4, 4] = inf_val
        for (ix, iy, v) in setters:
            h[ix, iy] = v
        if own is not None:
            h[4, 4] = own
        rec[name + "_h"] = h; rec[name + "_inf"] = inf; rec[name + "_dh"] = ref_guess(h, inf)
    gcase("G1", [(2, 4, 1.0)])
    gcase("G2", [(4, 2, 1.0)])
    gcase("G3", [(6, 4, 0.2)])
    gcase("G4", [(5, 4, 0.1), (3, 4, 0.2), (4, 5, 0.3), (4, 3, -0.4)])
    gcase("G5", [(5, 4, 0.1)], own=0.3)
    xy2 = 24
    h = rng.normal(0, 0.5, (xy2, xy2)); h[rng.uniform(size=(xy2, xy2)) < 0.93] = -1000.0
    inf = rng.normal(0, 0.5, (xy2, xy2)); inf[rng.uniform(size=(xy2, xy2)) < 0.3] = -1000.0
    rec["G6_h"] = h; rec["G6_inf"] = inf; rec["G6_dh"] = ref_guess(h, inf)
    save("kat_2d", rec)


def ref_positive(index_map, height, xy, zs, z_res, pos_thr, hit, total, robot_height, origin,
                 sx, sy, slope_thr):
    out = cuda.to_device(np.zeros((xy, xy), np.int32))
    kern("make_positive_obstacle_map")[_blocks2d(xy), (16, 16)](
        cuda.to_device(index_map), cuda.to_device(height), xy, zs, z_res, pos_thr,
        cuda.to_device(hit), cuda.to_device(total), robot_height,
        cuda.to_device(np.asarray(origin, np.float64)), cuda.to_device(sx), cuda.to_device(sy),
        slope_thr, out)
    return out.copy_to_host()


def kat_positive_and_fusion():
    rec = {}
    xy, zs = 4, 8
    V = xy * xy * zs

    def pcase(name, robot_height, vox, sx00=0.0):
        index_map = np.full(V, -1, np.int32)
        hit = []; total = []
        for (z, h, t) in vox:
            index_map[0 + 0 * xy + z * xy * xy] = len(hit)
            hit.append(h); total.append(t)
        hit = np.asarray(hit + [0], np.int32); total = np.asarray(total + [0], np.int32)
        height = np.full((xy, xy), -0.8)
        sx = np.zeros((xy, xy)); sy = np.zeros((xy, xy)); sx[0, 0] = sx00
        out = ref_positive(index_map, height, xy, zs, 0.2, 0.5, hit, total, robot_height,
                           (0.0, 0.0, -4.0), sx, sy, 0.3)
        rec[name + "_index_map"] = index_map; rec[name + "_hit"] = hit; rec[name + "_total"] = total
        rec[name + "_height"] = height; rec[name + "_sx"] = sx; rec[name + "_sy"] = sy
        rec[name + "_scal"] = np.asarray([xy, zs, 0.2, 0.5, robot_height, 0.3], np.float64)
        rec[name + "_origin"] = np.asarray((0.0, 0.0, -4.0))
        rec[name + "_out"] = out
    pcase("P1", 2.0, [])
    pcase("P2", 0.6, [(3, 29, 100), (4, 5, 50)])
    pcase("P3", 1.0, [(3, 29, 100), (4, 11, 20), (5, 12, 30)])
    pcase("P4a", 0.6, [], sx00=0.3)
    pcase("P4b", 0.6, [], sx00=0.2999)
    # D: __combine_old_indices transition table, one voxel per case in a 16x1... use 4x4x1 grid
    xy, zs = 4, 1
    before = np.asarray([-1, -2, -11, -12, -50, 3, -1, -2, -11, -12, 3, -1, -1, -1, -1, -1], np.int32)
    old = np.asarray([7, 7, 7, 7, 7, 7, -5, -5, -5, -5, -5, -1, -1, -1, -1, -1], np.int32)
    cnt = cuda.to_device(np.zeros(1, np.int64))
    comb = cuda.to_device(before.copy())
    o = cuda.to_device(np.zeros(3))
    kern("combine_old_indices")[(1, 1, 1), (8, 8, 4)](cnt, comb, o, cuda.to_device(old), 16, o, xy, zs)
    after = comb.copy_to_host()
    rec["D_before"] = before; rec["D_old"] = old; rec["D_after_occupied"] = (after >= 0)
    rec["D_after_free"] = np.where(after >= 0, 0, after); rec["D_count"] = cnt.copy_to_host()
    save("kat_positive_fusion", rec)


def main():
    names = sys.argv[1:] or ["kat", "f1", "f2", "f3", "f4", "f5", "f6"]
    for name in names:
        t0 = time.time()
        if name == "kat":
            kat_point_2_map(); kat_transform(); kat_2d(); kat_positive_and_fusion()
        else:
            sc = scenarios.SCENARIOS[name]()
            t_steps = []
            # time each step (BASELINE.md section 5 wants the c1 numbers)
            rec = scenarios.run_and_record(lambda *p: _Timed(gvom.Gvom(*p), t_steps), sc,
                                           record_debug=(name != "f7"))
            rec["ref_step_seconds"] = np.asarray(t_steps)
            save(name, rec)
        print("%s done in %.1f s" % (name, time.time() - t0))


class _Timed(object):
    """Transparent proxy that records wall time of process_pointcloud / combine_maps."""

    def __init__(self, g, sink):
        object.__setattr__(self, "_g", g)
        object.__setattr__(self, "_sink", sink)

    def __getattr__(self, k):
        return getattr(self._g, k)

    def process_pointcloud(self, *a, **kw):
        t = time.time(); r = self._g.process_pointcloud(*a, **kw); self._sink.append(time.time() - t)
        return r

    def combine_maps(self):
        t = time.time(); r = self._g.combine_maps(); self._sink.append(time.time() - t)
        return r


if __name__ == "__main__":
    main()

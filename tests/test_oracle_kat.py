"""Pins the CPU oracle (oracle/gvom_oracle.c) against kernel-level known-answer vectors
captured from the reference itself under Numba's CUDA simulator
(tests/golden/make_golden.py kat).  Integer outputs bit-exact; f64 maps to 1e-12."""
import os

import numpy as np
import pytest

from oracle import oracle

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_point_2_map_known_answers():
    rec = np.load(os.path.join(G, "kat_point_2_map.npz"))
    n = int(rec["n_cases"])
    assert n >= 20
    for k in range(n):
        pre = "c%d_" % k
        xy_res, z_res, xy, zs, md = rec[pre + "scal"]
        hit, total, _ = oracle.point_2_map(float(xy_res), float(z_res), int(xy), int(zs), float(md),
                                           rec[pre + "pts"], rec[pre + "ego"], rec[pre + "origin"])
        name = str(rec[pre + "name"])
        assert np.array_equal(hit, rec[pre + "hit"]), name
        assert np.array_equal(total, rec[pre + "total"]), name


def test_point_2_map_survey_table():
    """SURVEY.md Appendix C.3 rows spelled out by hand (independent of the .npz)."""
    def run(ego, pt, **kw):
        hit, total, _ = oracle.point_2_map(kw.get("xy_res", 1.0), kw.get("z_res", 1.0), 8, 4,
                                           kw.get("md", 0.0), np.asarray([pt], np.float64), ego,
                                           kw.get("origin", (0, 0, 0)))
        out = []
        for idx in np.nonzero(total)[0]:
            z, r = divmod(int(idx), 64); y, x = divmod(r, 8)
            out.append((x, y, z, int(hit[idx]), int(total[idx])))
        return sorted(out)
    e = (0.5, 0.5, 0.5)
    assert run(e, (5.5, .5, .5)) == [(1, 0, 0, 0, 1), (2, 0, 0, 0, 1), (3, 0, 0, 0, 1), (4, 0, 0, 0, 1), (5, 0, 0, 1, 1)]
    assert run(e, (5.9, .5, .5))[-1] == (5, 0, 0, 1, 2)
    assert run(e, (4.5, 4.5, .5)) == [(1, 1, 0, 0, 1), (2, 2, 0, 0, 1), (3, 3, 0, 0, 1), (4, 4, 0, 1, 2)]
    assert run(e, (20.5, .5, .5)) == [(i, 0, 0, 0, 1) for i in range(1, 8)]
    assert run(e, e) == [(0, 0, 0, 1, 1)]
    assert run(e, (1.2, .5, .5)) == [(1, 0, 0, 1, 1)]
    assert run((0, 0, 0), (2, 0, 0), md=3.0) == []
    assert run((2.5, 2.5, 3.5), (2.6, 2.7, 0.5)) == [(2, 2, 0, 1, 2), (2, 2, 1, 0, 1), (2, 2, 2, 0, 1)]
    assert run(e, (3.5, 3.5, 3.5)) == [(1, 1, 1, 0, 1), (2, 2, 2, 0, 1), (3, 3, 3, 1, 2)]
    assert run((.1, .1, .1), (1.0, .3, -.2), xy_res=.4, z_res=.2, origin=(-4, -4, -2), md=.5) == \
        [(5, 4, 1, 0, 1), (6, 4, 1, 1, 2)]


def test_transform_known_answers():
    rec = np.load(os.path.join(G, "kat_transform.npz"))
    for name in ("f32", "f64"):
        out = oracle.transform_pointcloud(rec[name + "_in"], rec["T"])
        assert out.dtype == rec[name + "_out"].dtype
        assert np.array_equal(out, rec[name + "_out"]), name


def test_slope_known_answers():
    rec = np.load(os.path.join(G, "kat_2d.npz"))
    for name in ("S1", "S2", "S3", "S4"):
        sx, sy, r = oracle.calculate_slope(rec[name + "_h"], 0.4)
        np.testing.assert_allclose(sx, rec[name + "_sx"], rtol=0, atol=1e-12, err_msg=name)
        np.testing.assert_allclose(sy, rec[name + "_sy"], rtol=0, atol=1e-12, err_msg=name)
        np.testing.assert_allclose(r, rec[name + "_r"], rtol=1e-12, atol=1e-12, err_msg=name)
    sx, sy, r = oracle.calculate_slope(rec["S1_h"], 0.4)
    assert sx[3, 3] == pytest.approx(0.24497866312686414, abs=1e-15)
    assert r[3, 3] == pytest.approx(-11.602141809704579, abs=1e-9)
    _, _, r2 = oracle.calculate_slope(rec["S2_h"], 0.4)
    assert r2[3, 3] == 0.0 and r2[4, 4] == 0.0 and r2[2, 2] == -1.0


def test_guess_height_known_answers():
    rec = np.load(os.path.join(G, "kat_2d.npz"))
    for name in ("G1", "G2", "G3", "G4", "G5", "G6"):
        dh = oracle.guess_height(rec[name + "_h"], rec[name + "_inf"])
        assert np.array_equal(dh, rec[name + "_dh"]), name
    assert oracle.guess_height(rec["G1_h"], rec["G1_inf"])[4, 4] == 1001.0      # typo B.6
    assert oracle.guess_height(rec["G2_h"], rec["G2_inf"])[4, 4] == 0.0


def test_positive_obstacle_known_answers():
    rec = np.load(os.path.join(G, "kat_positive_fusion.npz"))
    expect = {"P1": 0, "P2": 28, "P3": 34, "P4a": 100, "P4b": 0}
    for name, want in expect.items():
        xy, zs, z_res, pos_thr, robot_h, slope_thr = rec[name + "_scal"]
        out = oracle.make_positive_obstacle_map(rec[name + "_index_map"], rec[name + "_height"],
                                                int(xy), int(zs), float(z_res), float(pos_thr),
                                                rec[name + "_hit"], rec[name + "_total"],
                                                float(robot_h), rec[name + "_origin"],
                                                rec[name + "_sx"], rec[name + "_sy"], float(slope_thr))
        assert np.array_equal(out, rec[name + "_out"]), name
        assert out[0, 0] == want, name


def test_combine_old_indices_transition_table():
    rec = np.load(os.path.join(G, "kat_positive_fusion.npz"))
    after, cnt = oracle.combine_old_indices(rec["D_before"], rec["D_old"], 4, 1)
    assert np.array_equal(after >= 0, rec["D_after_occupied"])
    assert np.array_equal(np.where(after >= 0, 0, after), rec["D_after_free"])
    assert cnt == int(rec["D_count"][0])
    # SURVEY C.4 row D by hand
    want_free = [0, 0, 0, -12, -50, 0, -5, -6, -15, -16, 0]
    assert list(np.where(after >= 0, 0, after)[:11]) == want_free
    assert list((after >= 0)[:11]) == [True, True, True, False, False, True, False, False, False, False, True]


def test_ros_occupancy_grids_known_answers():
    """gvom_ros.py:141-165 restated (oracle.ros_occupancy_grids): thresholds, the '+ min' roughness
    rescale and numpy's wrapping float64 -> int8 cast, on a hand-computed 2x2 example."""
    from oracle import oracle
    obs = np.array([[0, 30], [50, 51]], np.int32)          # [x, y]
    neg = np.array([[100, 0], [0, 0]], np.int32)
    cert = np.array([[1, 0], [1, 1]], np.int32)
    rough = np.array([[-1.0, -3.2], [-25.0, 0.5]], np.float64)
    hard, soft, c, n, r = oracle.ros_occupancy_grids((None, obs, neg, rough, cert))
    # order='F': index = x + 2*y
    assert hard.tolist() == [100, 0, 0, 100]               # neg at (0,0); 51 > 50 at (1,1)
    assert soft.tolist() == [0, 100, 100, 0]               # 0 < obs <= 50
    assert c.tolist() == [100, 100, 0, 100]
    assert n.tolist() == [100, 0, 0, 0]
    # (clip(r,-10,0) + -10)/10*100 : -1 -> -110 ; -25 -> -200 -> wraps to 56 ;
    # -3.2 -> -131.99999999999997 in f64 -> truncates to -131 -> wraps to 125 ; 0.5 -> -100
    assert r.tolist() == [-110, 56, 125, -100]
    assert all(a.dtype == np.int8 for a in (hard, soft, c, n, r))

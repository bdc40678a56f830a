"""Host-side pieces of the ingest row (reference gvom_ros.py:93-109): the node's tf
matrix and the ros_numpy xyz extraction, both restated (third-party code that is not part of the
reference checkout).  No GPU needed."""
import numpy as np

import gvom
import scenarios
from oracle import oracle


def test_transform_from_translation_rotation_known_answers():
    th = 0.7
    m = gvom.transform_from_translation_rotation((1.0, -2.0, 0.5), (0.0, 0.0, np.sin(th / 2), np.cos(th / 2)))
    assert np.allclose(m, scenarios.rot_z(th, (1.0, -2.0, 0.5)), atol=1e-15)
    # unnormalised quaternions are normalised (q *= sqrt(2/q.q)); a zero quaternion gives identity
    m2 = gvom.transform_from_translation_rotation((0, 0, 0), (0.0, 0.0, 3 * np.sin(th / 2), 3 * np.cos(th / 2)))
    assert np.allclose(m2, scenarios.rot_z(th, (0, 0, 0)), atol=1e-15)
    assert np.array_equal(gvom.transform_from_translation_rotation((4, 5, 6), (0, 0, 0, 0))[:3, :3], np.eye(3))
    # 90 degrees about x: y -> z
    s = np.sqrt(0.5)
    m3 = gvom.transform_from_translation_rotation((0, 0, 0), (s, 0, 0, s))
    assert np.allclose(m3 @ np.array([0, 1, 0, 1.0]), [0, 0, 1, 1], atol=1e-15)


def test_pointcloud2_to_xyz_array_drops_non_finite_records_and_widens():
    rec = np.zeros(5, dtype=[("x", "<f4"), ("pad", "<u4"), ("y", "<f4"), ("z", "<f4"), ("ring", "<u2"), ("t", "<u2")])
    rec["x"] = [1.5, np.nan, 3.0, 4.0, 5.0]
    rec["y"] = [0.1, 0.2, np.inf, 0.4, 0.5]
    rec["z"] = [-1, -2, -3, -np.inf, 0.25]
    rec["pad"] = 0xdeadbeef
    out = oracle.pointcloud2_to_xyz_array(rec.tobytes(), 5, rec.dtype.itemsize, (0, 8, 12))
    assert out.dtype == np.float64 and out.shape == (2, 3)
    assert out.tolist() == [[1.5, float(np.float32(0.1)), -1.0], [5.0, 0.5, 0.25]]
    keep = oracle.pointcloud2_to_xyz_array(rec.tobytes(), 5, rec.dtype.itemsize, (0, 8, 12), remove_nans=False)
    assert keep.shape == (5, 3)

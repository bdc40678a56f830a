"""Shared comparison rules for parity tests (fixtures vs oracle vs HIP).

Tolerances (BASELINE.json north_star): integer/index outputs bit-exact; float height /
slope / roughness within 1e-5.  In practice everything except log/atan2 results is
bit-identical, so the float maps are held to 1e-9 where both sides use the same libm and
1e-5 only across libm implementations (glibc vs ocml).
"""
import numpy as np

INT_KEYS = ("positive", "negative", "visibility", "fused_state", "fused_hit", "fused_total",
            "slot_state", "slot_hit", "slot_total", "cell_count", "buffer_index",
            "last_buffer_index", "slots_filled", "slot", "returned_none", "occupancy", "kind")
EXACT_FLOAT_KEYS = ("fused_min_h", "slot_min_h", "slot_origin", "origin_world", "height_map",
                    "inferred_height_map", "guessed_height_delta")
TOL_FLOAT_KEYS = ("roughness", "roughness_map", "x_slope_map", "y_slope_map", "debug_height_map",
                  "debug_inferred_height_map")
# per-voxel statistics (float accumulation order is unspecified on a GPU; eigenvalue differences of
# nearly degenerate covariances are ill-conditioned): looser, absolute + relative
STATS_KEYS = ("debug_voxel_map",)
INPUT_KEYS = ("pc", "ego", "tf")


def compare_records(got, want, float_tol=1e-5, skip=(), stats_rtol=1e-4, stats_atol=1e-5):
    """Asserts that `got` reproduces every output recorded in `want`."""
    checked = 0
    for key in want.files if hasattr(want, "files") else want.keys():
        if key in ("params", "n_steps", "ref_step_seconds"):
            continue
        base = key.split("_", 1)[1] if key[0] == "s" and "_" in key else key
        if base in INPUT_KEYS or base in skip:
            continue
        if base in STATS_KEYS and key not in got:
            continue                                   # statistics are opt-in
        assert key in got, "missing output %s" % key
        a, b = np.asarray(got[key]), np.asarray(want[key])
        assert a.shape == b.shape, (key, a.shape, b.shape)
        if base in INT_KEYS:
            assert np.array_equal(a, b), "%s differs in %d places" % (key, int(np.sum(a != b)))
        elif base in EXACT_FLOAT_KEYS:
            assert np.array_equal(a, b), "%s differs, max |d|=%g" % (key, float(np.max(np.abs(a - b))))
        elif base in TOL_FLOAT_KEYS:
            assert a.dtype == b.dtype, (key, a.dtype, b.dtype)
            np.testing.assert_allclose(a, b, rtol=0, atol=float_tol, err_msg=key)
        elif base in STATS_KEYS:
            assert a.dtype == b.dtype, (key, a.dtype, b.dtype)
            np.testing.assert_array_equal(a[:, :3], b[:, :3], err_msg=key + " xyz")
            np.testing.assert_array_equal(a[:, 4], b[:, 4], err_msg=key + " hit")
            np.testing.assert_allclose(a[:, 3], b[:, 3], rtol=1e-6, atol=0, err_msg=key + " solid factor")
            np.testing.assert_allclose(a[:, 5:], b[:, 5:], rtol=stats_rtol, atol=stats_atol, err_msg=key + " eigen")
        else:
            raise AssertionError("no comparison rule for %s" % key)
        checked += 1
    return checked

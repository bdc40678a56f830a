#!/opt/conda/bin/python3.9
"""THIS CONTAINER ONLY.  Golden vectors for the step right behind the hot path: the post-processing the reference's ROS
node applies to combine_maps()'s 5-tuple before it publishes (reference gvom_ros.py:113-165).

The UNMODIFIED /root/reference/scripts/gvom_ros.py is imported with import-time stand-ins for the ROS packages this
container does not have (rospy, tf, tf2_ros, ros_numpy, nav_msgs.msg, sensor_msgs.msg, sensor_msgs.point_cloud2) and for
the `gvom` module (the node's post-processing is pure numpy on the tuple combine_maps returns, so the mapper is a
stand-in that hands out RECORDED tuples: the reference's own outputs of fixtures F3, F4, F5, plus one hand-made tuple
that walks the value ranges).  `VoxelMapper.cb_timer` then runs as written; every `publish(out_map)` is captured.
Nothing of the reference is copied: the fixture holds the input tuples and the published int8 arrays.

    /opt/conda/bin/python3.9 tests/golden/make_ros_golden.py      ->  tests/golden/ros_f3.npz
"""
import os
import sys
import types

import numpy as np

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference/scripts"

PARAM_SETS = {          # ROS private parameters that differ from the node's defaults (gvom_ros.py:23-41)
    "p0": {},                                                                        # density 50, roughness -10 .. 0
    "p1": {"~density_threshold": 12.5, "~min_roughness": -6.0, "~max_roughness": 1.5},
}


def install_stubs(param_overrides, published):
    class Pub(object):
        def __init__(self, topic, *a, **k):
            self.topic = topic

        def publish(self, msg):
            data = getattr(msg, "data", None)
            if isinstance(data, np.ndarray):
                published.append((self.topic, np.array(data, copy=True), float(msg.info.origin.position.x),
                                  float(msg.info.origin.position.y), float(msg.info.resolution), int(msg.info.width)))

    class NS(object):
        def __init__(self, **k):
            self.__dict__.update(k)

    rospy = types.ModuleType("rospy")
    rospy.get_param = lambda name, default=None: param_overrides.get(name, default)
    rospy.Subscriber = lambda *a, **k: None
    rospy.Publisher = Pub
    rospy.Timer = lambda *a, **k: None
    rospy.Duration = lambda x: x
    rospy.Time = NS(now=lambda: 0.0)
    rospy.loginfo = lambda *a, **k: None
    tf = types.ModuleType("tf")
    tf.TransformerROS = lambda *a, **k: NS()
    tf2 = types.ModuleType("tf2_ros")
    tf2.Buffer = lambda *a, **k: NS()
    tf2.TransformListener = lambda *a, **k: NS()
    rn = types.ModuleType("ros_numpy")
    rn.point_cloud2 = NS()

    class OccupancyGrid(object):
        def __init__(self):
            self.header = NS(stamp=None, frame_id=None)
            self.info = NS(resolution=None, width=None, height=None,
                           origin=NS(orientation=NS(x=0, y=0, z=0, w=1), position=NS(x=0, y=0, z=0)))
            self.data = None
    nav = types.ModuleType("nav_msgs")
    nav_msg = types.ModuleType("nav_msgs.msg")
    nav_msg.Odometry = NS
    nav_msg.OccupancyGrid = OccupancyGrid
    nav.msg = nav_msg
    sm = types.ModuleType("sensor_msgs")
    sm_msg = types.ModuleType("sensor_msgs.msg")
    sm_msg.PointCloud2 = NS
    sm_pc2 = types.ModuleType("sensor_msgs.point_cloud2")
    sm.msg, sm.point_cloud2 = sm_msg, sm_pc2

    class StandInGvom(object):            # the mapper: hands out recorded combine_maps() tuples
        def __init__(self, *params):
            self.params = params
            self.queue = []

        def combine_maps(self):
            return self.queue.pop(0)

        def make_debug_voxel_map(self):
            return None

        def make_debug_height_map(self):
            return None

        def make_debug_inferred_height_map(self):
            return None
    gv = types.ModuleType("gvom")
    gv.Gvom = StandInGvom
    for name, mod in (("rospy", rospy), ("tf", tf), ("tf2_ros", tf2), ("ros_numpy", rn), ("nav_msgs", nav),
                      ("nav_msgs.msg", nav_msg), ("sensor_msgs", sm), ("sensor_msgs.msg", sm_msg),
                      ("sensor_msgs.point_cloud2", sm_pc2), ("gvom", gv)):
        sys.modules[name] = mod


def recorded_tuples():
    """(tag, 5-tuple) of every combine of F3, F4, F5 + one hand-made tuple over the value ranges"""
    out = []
    for name in ("f3", "f4", "f5"):
        d = np.load(os.path.join(HERE, name + ".npz"))
        for k in range(int(d["n_steps"])):
            if int(d["s%d_kind" % k]) == 1 and not bool(d["s%d_returned_none" % k]):      # 1 = combine
                out.append(("%s_s%d" % (name, k), tuple(d["s%d_%s" % (k, f)] for f in
                                                         ("origin_world", "positive", "negative", "roughness", "visibility"))))
    n = 12
    pos = (np.arange(n * n, dtype=np.int32).reshape(n, n) * 7) % 101                   # 0 .. 100, both sides of every threshold
    neg = np.where((np.arange(n * n).reshape(n, n) % 5) == 0, 100, 0).astype(np.int32)
    vis = ((np.arange(n * n).reshape(n, n) % 3) != 0).astype(np.int32)
    rough = np.linspace(-80.0, 6.0, n * n).reshape(n, n)                                # log-residuals, the 0.0 and -1.0 defaults, above 0
    rough[0, :6] = [0.0, -1.0, -10.0, -6.0, 1.5, -0.0]
    out.append(("ranges", (np.array([-2.4, 3.2, -1.0]), pos, neg, rough, vis)))
    return out


def main():
    rec = {"numpy_version": np.__version__, "param_sets": np.array(sorted(PARAM_SETS))}
    tuples = recorded_tuples()
    for pname, overrides in sorted(PARAM_SETS.items()):
        published = []
        for m in [m for m in sys.modules if m == "gvom_ros"]:
            del sys.modules[m]
        n = tuples[0][1][1].shape[0]
        install_stubs(dict(overrides), published)
        if REF not in sys.path:
            sys.path.insert(0, REF)
        import gvom_ros                                                  # the reference's node, unmodified
        assert os.path.dirname(os.path.abspath(gvom_ros.__file__)) == REF
        node = gvom_ros.VoxelMapper()
        rec[pname + "_density_threshold"] = float(node.density_threshold)
        rec[pname + "_min_roughness"] = float(node.min_roughness)
        rec[pname + "_max_roughness"] = float(node.max_roughness)
        for tag, tup in tuples:
            node.width = tup[1].shape[0]
            node.voxel_mapper.queue.append(tuple(np.array(a, copy=True) for a in tup))
            del published[:]
            node.cb_timer(None)
            topics = [p[0] for p in published]
            assert topics == ["~hard_obstacle_map", "~soft_obstacle_map", "~ground_certainty_map", "~all_ground_certainty_map",
                              "~negative_obstacle_map", "~roughness_map"], topics
            for (topic, data, ox, oy, res, width), short in zip(published, ("hard", "soft", "certainty", "all_certainty", "negative", "roughness")):
                assert data.dtype == np.int8 and data.shape == (tup[1].size,)
                rec["%s_%s_%s" % (pname, tag, short)] = data
            assert np.array_equal(published[2][1], published[3][1])
            rec["%s_%s_origin_xy" % (pname, tag)] = np.array([published[0][2], published[0][3]])
    for tag, tup in tuples:
        for f, a in zip(("origin_world", "positive", "negative", "roughness", "visibility"), tup):
            rec["in_%s_%s" % (tag, f)] = np.asarray(a)
    rec["tags"] = np.array([t for t, _ in tuples])
    np.savez_compressed(os.path.join(HERE, "ros_f3.npz"), **rec)
    print("wrote ros_f3.npz: %d ticks x %d parameter sets, numpy %s" % (len(tuples), len(PARAM_SETS), np.__version__))


if __name__ == "__main__":
    main()

import os, sys, time, gc
os.environ.setdefault("GVOM_ENV_DYNAMIC", "1")
ROOT = "/root/repo" if os.path.exists("/root/repo/g-vom_amd") else os.environ.get("GRAFT_REPO_ROOT", ".")
sys.path.insert(0, os.path.join(ROOT, "g-vom_amd")); sys.path.insert(0, ROOT)
import numpy as np
import gvom, synth, bench
hip = bench.Hip(); hip.set_device(0)
for name, xy, zs, nsens in (("c4", 512, 128, 4), ("c5", 1024, 128, 16)):
    params = (0.2, 0.2, xy, zs, 1) + synth.REF_TAIL
    scene = synth.make_scene(2, extent=0.2 * xy / 2 * 0.9)
    ego = (0.0, 0.0, 0.0)
    pc = np.concatenate([synth.lidar_scan(scene, beams=128, sensor=ego, yaw=2 * np.pi / 2048 * r / nsens, noise_seed=r) for r in range(nsens)], 0)
    g = gvom.Gvom(*params, device=0); d = hip.to_device(pc)
    g.set_profiling(True)
    for segs in (6, 4, 3, 2, 1):
        os.environ["GVOM_TRACE_SEGMENTS"] = str(segs)
        acc = []
        for k in range(14):
            g.process_pointcloud_device(d.value, pc.shape[0], np.float32, ego, None); g.combine_maps()
            if k >= 4: acc.append(g.last_stage_ms()["trace"] * 1e3)
        print(name, "segments", segs, "trace us %.1f" % np.median(acc))
    del g

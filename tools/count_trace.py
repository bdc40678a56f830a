import sys, ctypes
sys.path.insert(0, "/root/repo/g-vom_amd"); sys.path.insert(0, "/root/repo")
import numpy as np, gvom, synth
for name in ("m256", "c2"):
    params, scans = synth.config_inputs(name)
    g = gvom.Gvom(*params)
    pc, ego, tf = scans[0]
    g.process_pointcloud(pc, ego)
    out = (ctypes.c_uint32 * 3)()
    g._lib.gvom_debug_trace_counters.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
    g._lib.gvom_debug_trace_counters(g._h, out)
    st = g.scan_stats()
    d = g.read_dense(0)
    print(name, "updates", st["sum_total"], "heads", out[0], "line-heads", out[1], "atomic wave-instr", out[2], "distinct voxels", int((d[0] != -1).sum()))

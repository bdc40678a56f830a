import os, sys, time, numpy as np
R=os.environ.get("GRAFT_REPO_ROOT","/root/repo"); sys.path[:0]=[R,R+"/g-vom_amd"]
import bench, gvom, synth
hip=bench.Hip(); hip.set_device(0)
params, scans = synth.config_inputs("c1")
pc, ego, tf = scans[0]
def dirkey(pc, ego, nb):
    d = pc/np.array([params[0],params[0],params[1]]) - np.array(ego)/np.array([params[0],params[0],params[1]])
    a = np.abs(d); ax = np.argmax(a,axis=1); idx=np.arange(len(d))
    s = d[idx,ax] < 0
    u = d[idx,(ax+1)%3]/a[idx,ax]; v = d[idx,(ax+2)%3]/a[idx,ax]
    qu = np.clip(((u+1)*0.5*nb).astype(int),0,nb-1); qv = np.clip(((v+1)*0.5*nb).astype(int),0,nb-1)
    # morton
    def part(x):
        x = (x | (x<<8)) & 0x00FF00FF; x=(x|(x<<4))&0x0F0F0F0F; x=(x|(x<<2))&0x33333333; x=(x|(x<<1))&0x55555555; return x
    m = part(qu) | (part(qv)<<1)
    return (ax*2+s)*nb*nb*4 + m
def run(cloud, label):
    g=gvom.Gvom(*params); d=hip.to_device(cloud)
    for k in range(60): g.process_pointcloud_device(d.value, cloud.shape[0], cloud.dtype, ego, tf); g.combine_maps()
    t0=time.perf_counter()
    for k in range(400): g.process_pointcloud_device(d.value, cloud.shape[0], cloud.dtype, ego, tf); g.combine_maps()
    us=(time.perf_counter()-t0)/400*1e6
    g.set_profiling(True); acc=[]
    for k in range(40): g.process_pointcloud_device(d.value, cloud.shape[0], cloud.dtype, ego, tf); g.combine_maps(); acc.append(g.last_stage_ms()["trace"])
    print(label, "%.1f us/step, trace %.1f us" % (us, float(np.median(acc))*1e3), flush=True)
run(pc, "c1 as given (uniform random order)")
for nb in (4, 8, 16, 32):
    k = dirkey(pc.astype(np.float64), ego, nb); o = np.argsort(k, kind="stable")
    run(np.ascontiguousarray(pc[o]), "c1 sorted by direction, %d x %d bins per cube face" % (nb, nb))

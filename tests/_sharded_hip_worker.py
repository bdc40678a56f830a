"""Worker of tests/test_hip_sharded.py: `world` ranks share cuda:0 (gloo stages the collectives
through the host), each owning one slab through the slab-sharded HIP handle; the result must
equal the single-handle HIP result on the concatenated cloud, bit for bit."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden"), os.path.join(ROOT, "g-vom_amd")):
    sys.path.insert(0, p)
import numpy as np
import torch.distributed as dist

import gvom
import gvom_sharded
import synth


def main():
    dist.init_process_group("gloo", init_method="env://")
    rank, world = dist.get_rank(), dist.get_world_size()
    params = (0.2, 0.2, 128, 32, 3, 1.0, 0.5, 0.5, 0.3, 2.0, 4.0, 1.0, 1, 1)
    sh = gvom_sharded.ShardedGvom(*params, device=0)
    ref = gvom.Gvom(*params, device=0) if rank == 0 else None
    scene = synth.make_scene(2, extent=11.0)
    assert sh.combine_maps() is None
    for k in range(5):
        ego = (0.7 * k, -0.45 * k, 0.05 * k)
        shares = [synth.lidar_scan(scene, beams=16, azimuths=1024, sensor=ego, yaw=0.001 * r, noise_seed=10 * k + r)
                  for r in range(world)]
        if k == 3:
            shares = [s + 900.0 for s in shares]                 # globally rejected scan
        sh.process_pointcloud(shares[rank], ego)
        got = sh.combine_maps()
        cnt = sh.combined_cell_count_cpu          # collective-backed: every rank must ask
        if rank == 0:
            ref.process_pointcloud(np.concatenate(shares, 0), ego)
            want = ref.combine_maps()
            for a, b in zip(got, want):
                assert a.dtype == b.dtype and np.array_equal(a, b), "step %d" % k
            assert cnt == ref.combined_cell_count_cpu, (k, cnt, ref.combined_cell_count_cpu)
    dist.barrier()
    dist.destroy_process_group()
    print("rank %d ok" % rank)


if __name__ == "__main__":
    main()

"""Worker of tests/test_sharded_gloo.py (one process per rank, gloo, CPU)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden"), os.path.join(ROOT, "g-vom_amd")):
    sys.path.insert(0, p)
import numpy as np
import torch.distributed as dist

import gvom_sharded
from shard_fake import OracleShardBackend, GlooComm
from oracle import oracle


def main():
    dist.init_process_group("gloo", init_method="env://")
    rank, world = dist.get_rank(), dist.get_world_size()
    params = (0.4, 0.2, 24, 12, 2, 0.5, 0.5, 0.5, 0.3, 2.0, 2.0, 1.0, 1, 1)
    sh = gvom_sharded.ShardedGvom(*params, comm=GlooComm(), backend=OracleShardBackend(params, rank, world))
    ref = oracle.OracleGvom(*params)
    assert sh.combine_maps() is None
    rng = np.random.default_rng(5)
    n_per_rank = 300
    for k in range(4):
        ego = (0.9 * k, -0.5 * k, 0.1 * k)                     # origin moves: storage offsets change
        full = np.stack([rng.uniform(-4, 4, world * n_per_rank) + ego[0],
                         rng.uniform(-4, 4, world * n_per_rank) + ego[1],
                         rng.normal(-0.6, 0.4, world * n_per_rank)], axis=1)
        if k == 2:
            full = full + 500.0                                 # no overlap: must be rejected globally
        # ragged shares: rank 0 gets a third, the last rank the rest; in step 1 rank 0's share is empty
        cut = [0] + [world * n_per_rank // 3 * (r + 1) // world for r in range(world - 1)] + [world * n_per_rank]
        if k == 1:
            cut[1] = 0
        share = full[cut[rank]:cut[rank + 1]]
        sh.process_pointcloud(share, ego)
        ref.process_pointcloud(full, ego)
        got, want = sh.combine_maps(), ref.combine_maps()
        assert (got is None) == (want is None)
        for a, b in zip(got, want):
            assert a.dtype == b.dtype and a.shape == b.shape
            assert np.array_equal(a, b), "step %d rank %d" % (k, rank)
        assert sh.combined_cell_count_cpu == ref.combined_cell_count_cpu
    dist.barrier()
    dist.destroy_process_group()
    print("rank %d ok" % rank)


if __name__ == "__main__":
    main()

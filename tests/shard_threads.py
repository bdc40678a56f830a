"""Runs `world` ranks of g-vom_amd/gvom_sharded.ShardedGvom as THREADS of this process on ONE GPU
(gvom_sharded.ThreadComm: the handles' own exchange regions, moved with hipMemcpyAsync on the receiving handle's stream, in the order,
sizes and stream ordering of the RCCL path).  Used by the GPU tests: real pack / unpack / slab kernels, real split
C-ABI entry points, SPMD orchestration -- only the wire is not xGMI."""
import threading

import gvom_sharded


def run_ranks(world, params, body, device=0, **kw):
    """body(rank, sharded_gvom) -> result; returns [result per rank]; re-raises the first failure.
    kw: further keyword arguments of ShardedGvom (voxel_statistics=True, ...)."""
    fabric = gvom_sharded.ThreadFabric(world)
    results, errors = [None] * world, [None] * world

    def worker(r):
        try:
            sh = gvom_sharded.ShardedGvom(*params, comm=fabric.comm(r), device=device, **kw)
            results[r] = body(r, sh)
        except BaseException as e:          # noqa: BLE001 -- reported to the caller below
            errors[r] = e
            fabric.barrier.abort()

    threads = [threading.Thread(target=worker, args=(r,)) for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    for e in errors:
        if e is not None and not isinstance(e, threading.BrokenBarrierError):
            raise e
    for e in errors:
        if e is not None:
            raise e
    return results

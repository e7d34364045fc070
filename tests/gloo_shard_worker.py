"""Worker of the two-rank tests: one rank of a 2-rank sharded batch, rendezvous over gloo.

CPU tests (tests/test_host.py): the device solve is replaced by the oracle (a test may use it); what is
exercised is the product's sharding code: the deterministic LPT partition, the row gather and the MAX
reduction.  `device` as third argument (tests/test_gpu_multirank.py, -m gpu): every rank solves its shard with
the HIP path - its own pantr_batch on the one GPU the box has, the two processes sharing it - and rank 0
compares the gathered result with the unsharded device solve and with the oracle.
"""
import os
import sys
import time

import numpy as np
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "pantr-cdc2023-experiments_amd"))
import oracle_binding as ob  # noqa: E402
from pantr_amd import sharding  # noqa: E402


def main():
    dist.init_process_group(backend="gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    if len(sys.argv) > 2 and sys.argv[2] == "fail":
        # rank 1 fails a local check: both ranks must leave together, before any further collective
        def check():
            if rank == 1:
                raise SystemExit("worker: rank 1 failed its check")
        sharding.raise_together(sharding.local_failure(check))
        dist.barrier()  # never reached
        return
    N, B = 6, 7
    rng = np.random.default_rng(5)
    x = ob.initial_state()
    for _ in range(5):
        x = ob.simulate(x, (-1.0, 1.0, 1.0))
    x0 = x[None, :] + 0.02 * rng.standard_normal((B, x.size))
    u0 = rng.uniform(-0.2, 0.2, (B, 3 * N))
    u0[::2] = 0.0                                   # "cold" instances
    costs = np.where(np.all(u0 == 0, axis=1), 2.0, 1.0)
    idx = sharding.shard_indices(B, rank, world, costs)
    device = len(sys.argv) > 2 and sys.argv[2] == "device"
    t = time.perf_counter()
    if device:  # the product path: this rank's own handle on the shared GPU
        import pantr_amd as pa
        batch = pa.PantrBatch("hanging_chain", N, max(len(idx), 1), device=0, opts=pa.DODO_PANTR_OPTS)
        us = batch.solve(x0[idx], u0[idx])[0] if len(idx) else np.zeros((0, 3 * N))
    else:
        us, _ = ob.solve_batch(N, x0[idx], u0[idx]) if len(idx) else (np.zeros((0, 3 * N)), [])
    elapsed = time.perf_counter() - t
    gathered = sharding.gather_rows(us, idx, B)
    tmax = sharding.max_over_ranks(elapsed)
    times = [None] * world
    dist.all_gather_object(times, elapsed)
    idxs = [None] * world
    dist.all_gather_object(idxs, idx)
    if rank == 0:
        serial, _ = ob.solve_batch(N, x0, u0)
        extra = {}
        if device:
            full = pa.PantrBatch("hanging_chain", N, B, device=0, opts=pa.DODO_PANTR_OPTS)
            extra["unsharded_device"] = full.solve(x0, u0)[0]
            extra["kernel"] = np.array(batch.kernel_name(max(len(idx), 1)))
            full.close()
        np.savez(sys.argv[1], gathered=gathered, serial=serial, tmax=tmax, t0=times[0], t1=times[1],
                 idx0=idxs[0], idx1=idxs[1], world=world, **extra)
    if device:
        batch.close()
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()

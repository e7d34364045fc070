"""Sharding of independent MPC instances over the GPUs of one node (one process per GPU).

The path has no exchange step: instances are dealt to ranks on the
host, every rank solves its shard with its own pantr_batch, and ranks only meet to reduce the
elapsed time (MAX) and, when a caller wants the full result on every rank, to gather
solutions.  torch.distributed supplies those two collectives (RCCL on GPUs, gloo in the CPU
tests); nothing on the data path of a solve is collective.

The batch is what new-benchmarks-paper/dodo.py:69-111 enumerates (problem x {warm, cold} x
horizon x time step); the expected cost of an instance is known roughly in advance (cold
solves and long horizons are slower), hence longest-processing-time-first dealing.
"""
import numpy as np


def lpt_partition(costs, world):
    """Deal instances to `world` ranks, most expensive first, always to the least loaded rank.

    Deterministic (ties broken by instance index, then by rank), so every rank computes the
    same partition without communicating.  Returns a list of `world` sorted index arrays.
    """
    costs = np.asarray(costs, dtype=np.float64)
    order = np.lexsort((np.arange(len(costs)), -costs))
    load = np.zeros(world)
    shards = [[] for _ in range(world)]
    for i in order:
        r = int(np.argmin(load))  # first minimum: lowest rank on ties
        shards[r].append(int(i))
        load[r] += costs[i]
    return [np.array(sorted(s), dtype=np.int64) for s in shards]


def shard_indices(B, rank, world, costs=None):
    """Indices of the instances rank `rank` solves (uniform costs: round-robin)."""
    if costs is None:
        return np.arange(rank, B, world, dtype=np.int64)
    return lpt_partition(costs, world)[rank]


def launch_order(costs):
    """Workgroup start order inside one GPU: most expensive instance first (shortens the tail)."""
    costs = np.asarray(costs, dtype=np.float64)
    return np.lexsort((np.arange(len(costs)), -costs)).astype(np.int32)


def max_over_ranks(value, device=None):
    """MAX-reduce a host float over all ranks (identity when not distributed)."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return float(value)
    t = torch.tensor([value], dtype=torch.float64, device=device or "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def raise_together(message, device=None):
    """Called by EVERY rank at the same point with its local failure message (None = this rank is fine):
    if any rank failed, all ranks raise SystemExit here - the failing ones with their own message -
    instead of one rank leaving while its peers walk into the next collective and wait for its timeout."""
    if max_over_ranks(1.0 if message else 0.0, device) > 0:
        raise SystemExit(message or "another rank reported a failure (see its message); leaving with it")


def local_failure(fn, *args):
    """fn(*args), returning the message of the SystemExit it raises (None if it does not)."""
    try:
        fn(*args)
    except SystemExit as e:
        return str(e.code) if e.code is not None else "failed"
    return None


def gather_rows(local_rows, local_idx, total, device=None):
    """All-gather per-instance rows (solutions, stats columns) of a sharded batch.

    local_rows: (len(local_idx), k) float64 array of this rank; returns the (total, k) array
    assembled on every rank.  Shards may have different sizes (padded for the collective).
    """
    import torch
    import torch.distributed as dist
    local_rows = np.ascontiguousarray(local_rows, dtype=np.float64)
    k = local_rows.shape[1]
    out = np.zeros((total, k))
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        out[np.asarray(local_idx)] = local_rows
        return out
    world = dist.get_world_size()
    dev = device or "cpu"
    cnt = torch.tensor([len(local_idx)], dtype=torch.int64, device=dev)
    counts = [torch.zeros_like(cnt) for _ in range(world)]
    dist.all_gather(counts, cnt)
    cap = int(max(int(c.item()) for c in counts))
    pad_rows = torch.zeros((cap, k), dtype=torch.float64, device=dev)
    pad_idx = torch.full((cap,), -1, dtype=torch.int64, device=dev)
    if len(local_idx):
        pad_rows[:len(local_idx)] = torch.from_numpy(local_rows).to(dev)
        pad_idx[:len(local_idx)] = torch.from_numpy(np.asarray(local_idx, dtype=np.int64)).to(dev)
    rows = [torch.zeros_like(pad_rows) for _ in range(world)]
    idxs = [torch.zeros_like(pad_idx) for _ in range(world)]
    dist.all_gather(rows, pad_rows)
    dist.all_gather(idxs, pad_idx)
    for r, i in zip(rows, idxs):
        i = i.cpu().numpy()
        ok = i >= 0
        out[i[ok]] = r.cpu().numpy()[ok]
    return out

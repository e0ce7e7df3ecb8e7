"""Batch sharding across ranks (one process per GPU). The batch axis of independent LQR problems
is the only thing that is ever split: rank r of W owns global problems [r*B, (r+1)*B) of a job of
W*B problems (weak scaling, B = per-GPU batch). There is no data-path collective; the helpers
below are the control-plane pieces bench.py and the tests share (barrier, max-over-ranks,
optional gather of solutions to rank 0). Backend "nccl" is RCCL on ROCm; "gloo" is used by the
CPU tests."""
import os

import numpy as np


def env_rank():
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")),
            int(os.environ.get("WORLD_SIZE", "1")))


def shard_range(rank, world, batch_per_rank):
    """Global problem indices owned by `rank`."""
    lo = rank * batch_per_rank
    return lo, lo + batch_per_rank


def shard_seed0(rank, batch_per_rank, job_seed0=1):
    """Seed of the first problem of the shard: global problem g has seed job_seed0 + g
    (SURVEY.md 8d), whatever the number of ranks."""
    return job_seed0 + rank * batch_per_rank


def max_over_ranks(values, device=None):
    """Element-wise MAX of a small list of floats over all ranks (identity when not distributed)."""
    import torch
    import torch.distributed as dist
    t = torch.tensor(list(values), dtype=torch.float64, device=device)
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return [float(v) for v in t.tolist()]


def all_over_ranks(value, device=None):
    """One float per rank -> the list of every rank's value, in rank order (identity when not distributed)."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return [float(value)]
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    parts = [torch.empty_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(parts, t)
    return [float(p.item()) for p in parts]


def gather_solutions(local, device=None):
    """all_gather of per-rank solution arrays [B, nvars] -> [W*B, nvars] in global problem order."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return np.asarray(local)
    t = torch.as_tensor(np.ascontiguousarray(local), dtype=torch.float64, device=device)
    parts = [torch.empty_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(parts, t)
    return torch.cat(parts, dim=0).cpu().numpy()


def timed_region(solver, steps, warmup, barrier):
    """bench.py's timed region, shared with the CPU rehearsal (tests/test_sharding_gloo.py):
    `warmup` untimed steps, then exactly `steps` steps bracketed by barrier() on both sides.
    `solver` needs solve_async() and synchronize(). Returns the rank's elapsed seconds."""
    import time
    for _ in range(warmup):
        solver.solve_async()
    solver.synchronize()
    barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        solver.solve_async()
    solver.synchronize()
    barrier()
    return time.perf_counter() - t0


def timed_region_with_gather(solver, steps, barrier, gather):
    """The same steps, each followed by the collection of every shard's solutions (SURVEY.md 8(e):
    throughput including the gather). gather() -> gathered array or None. Returns (seconds, last)."""
    import time
    barrier()
    t0 = time.perf_counter()
    last = None
    for _ in range(steps):
        solver.solve_async()
        solver.synchronize()
        last = gather()
    barrier()
    return time.perf_counter() - t0, last


# ------------------------------------------------------------------------------------------------ time axis
def solve_time_sharded(solver, rank, world, reduce_sum=None):
    """One solve of a problem whose HORIZON is cut over `world` ranks (include/ndlqr_hip.h): rank
    `rank` factors the tree levels inside its chunk of N / world knots, the world - 1 accumulator slots between the
    chunks are summed over the ranks (`reduce_sum(array)` -> summed array, in place or returned; default: all_reduce
    of a host tensor over the initialised process group), every rank eliminates the top log2(world) levels itself and
    back-substitutes its chunk. Returns the solver's status; afterwards solver.solutions() holds valid knots
    [rank N / world, (rank + 1) N / world)."""
    count = solver.time_shard_top_doubles(world)
    if count <= 0:
        return -1
    err = solver.time_shard_factor(rank, world)
    if err:
        return err
    buf = np.zeros(count)
    err = solver.time_shard_export(world, buf.ctypes.data)
    if err:
        return err
    if reduce_sum is None:
        import torch
        import torch.distributed as dist
        t = torch.from_numpy(buf)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
    else:
        out = reduce_sum(buf)
        if out is not None:
            buf = np.ascontiguousarray(out)
    err = solver.time_shard_import(world, buf.ctypes.data)
    if err:
        return err
    err = solver.time_shard_finish(rank, world)
    return err or solver.synchronize()


def chunk_of_solution(sol, n, m, N, rank, world):
    """The entries of packed solution vectors [.., nvars] that rank `rank`'s chunk of the horizon owns."""
    zb = 2 * n + m
    lo, hi = rank * (N // world) * zb, min((rank + 1) * (N // world) * zb, zb * N - m)
    return slice(lo, hi)

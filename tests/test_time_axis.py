"""Time-axis sharding: ONE problem's horizon cut over G ranks. Each rank factors the tree levels inside
its chunk, the G - 1 accumulator slots between the chunks are summed over the ranks (the only exchange of a solve),
every rank eliminates the top log2(G) levels itself and back-substitutes its chunk. The loop being split is the
reference's level loop, src/solve.c:68-134 (factor) and :137-182 (solve); the result must be the reference's."""
import os
import socket
import sys

import numpy as np
import pytest

from support import Problem

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
REL_TOL = 1e-9


def _problem(ndlqr, n, m, N, seed):
    g = ndlqr.generate_synthetic(n, m, N, seed)
    return g, Problem(n, m, N, g["A"], g["B"], g["Q"], g["R"], g["q"], g["r"], g["d"], g["x0"])


@pytest.mark.parametrize("n,m,N,batch,G", [(12, 4, 4096, 1, 2), (12, 4, 512, 3, 2), (12, 4, 1024, 2, 4), (6, 3, 256, 2, 8),
                                           (13, 4, 128, 1, 2), (12, 4, 8192, 1, 2)])
def test_time_axis_chunks_in_one_process(ndlqr, oracle, n, m, N, batch, G):
    """The G chunks as G solvers of one process, the sum of the top slots formed on the host: the algorithm without a
    process group. Every chunk's knots against the oracle, and the assembled vector against the plain solve."""
    from rslqr_amd import sharding
    gens, probs = zip(*[_problem(ndlqr, n, m, N, 40 + p) for p in range(batch)])
    flat = [np.stack([g[k] for g in gens]) for k in ("A", "B", "Q", "R", "q", "r", "d", "x0")]
    solvers = [ndlqr.BatchSolver(n, m, N, batch) for _ in range(G)]
    for bs in solvers:
        bs.initialize_flat(*flat)
    count = solvers[0].time_shard_top_doubles(G)
    assert count > 0
    for rep in range(2):  # (a second solve: nothing of the first exchange may be left in the slots)
        bufs = []
        for g, bs in enumerate(solvers):
            assert bs.time_shard_factor(g, G) == 0
            buf = np.zeros(count)
            assert bs.time_shard_export(G, buf.ctypes.data) == 0
            bufs.append(buf)
        total = np.sum(bufs, axis=0)
        for g, bs in enumerate(solvers):
            assert bs.time_shard_import(G, total.ctypes.data) == 0
            assert bs.time_shard_finish(g, G) == 0
            assert bs.synchronize() == 0 and bs.cholesky_failures() == 0
            assert bs.schedule() == "reduced-time-shard"
        whole = np.zeros((batch, solvers[0].nvars))
        for g, bs in enumerate(solvers):
            sl = sharding.chunk_of_solution(None, n, m, N, g, G)
            whole[:, sl] = bs.solutions()[:, sl]
        for p in sorted({0, batch - 1}):
            ref = oracle.solve(probs[p], 8)[0][: probs[p].nvars]
            assert np.linalg.norm(whole[p] - ref) / np.linalg.norm(ref) <= REL_TOL, (rep, p)
            res, bn = oracle.kkt_residual(probs[p], whole[p])
            assert res <= 1e-9 * max(1.0, bn)
    plain = ndlqr.BatchSolver(n, m, N, batch)
    plain.initialize_flat(*flat)
    assert plain.solve() == 0
    assert np.linalg.norm(plain.solutions() - whole) <= 1e-11 * np.linalg.norm(whole)
    plain.close()
    # the solvers go back to ordinary solves afterwards
    assert solvers[0].solve() == 0
    assert np.linalg.norm(solvers[0].solutions() - whole) <= 1e-11 * np.linalg.norm(whole)
    for bs in solvers:
        bs.close()


def test_time_axis_refusals(ndlqr):
    bs = ndlqr.BatchSolver(12, 4, 64, 1)
    bs.initialize_synthetic(1)
    assert bs.time_shard_top_doubles(3) < 0          # not a power of two
    assert bs.time_shard_factor(0, 8) != 0           # chunks shorter than 16 knots
    assert bs.time_shard_factor(2, 2) != 0           # rank out of range
    bs.close()
    bs = ndlqr.BatchSolver(20, 6, 256, 1)            # no size-specialised instance
    assert bs.time_shard_top_doubles(2) < 0
    bs.close()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _rank_main(rank, world, port, n, m, N, outdir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    sys.path.insert(0, HERE)
    sys.path.insert(0, os.path.dirname(HERE))
    import torch
    import torch.distributed as dist
    import rslqr_amd
    from rslqr_amd import sharding
    dist.init_process_group("gloo", rank=rank, world_size=world)
    bs = rslqr_amd.BatchSolver(n, m, N, 1, device=0)  # both ranks on the one GPU of the test box
    bs.initialize_synthetic(71)
    for _ in range(2):
        assert sharding.solve_time_sharded(bs, rank, world) == 0
    sl = sharding.chunk_of_solution(None, n, m, N, rank, world)
    mine = np.zeros(bs.nvars)
    mine[sl] = bs.solutions()[0, sl]
    t = torch.from_numpy(mine)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)  # chunks are disjoint: the sum is the assembled vector
    if rank == 0:
        np.save(os.path.join(outdir, "whole.npy"), mine)
    bs.close()
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,N", [(2, 4096), (4, 2048)])
def test_time_axis_rank_processes(ndlqr, oracle, tmp_path, world, N):
    """Rank PROCESSES (gloo; all on the one GPU of the test box -- with a GPU each the same code runs over RCCL),
    rslqr_amd.sharding.solve_time_sharded: batch 1, (12,4), N = 4096 over two ranks, N = 2048 over four (the top TWO
    levels eliminated redundantly by every rank, three slots in the all-reduce)."""
    import torch.multiprocessing as mp
    n, m = 12, 4
    mp.spawn(_rank_main, args=(world, _free_port(), n, m, N, str(tmp_path)), nprocs=world, join=True)
    whole = np.load(tmp_path / "whole.npy")
    _, prob = _problem(ndlqr, n, m, N, 71)
    ref = oracle.solve(prob, 8)[0][: prob.nvars]
    assert np.linalg.norm(whole - ref) / np.linalg.norm(ref) <= REL_TOL
    res, bn = oracle.kkt_residual(prob, whole)
    assert res <= 1e-9 * max(1.0, bn)

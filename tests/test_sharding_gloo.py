"""N>1 path on CPU: world_size-2 gloo. The batch is sharded by rank with global seeds and driven
through the SAME control flow bench.py runs on the GPUs (rslqr_amd.sharding.timed_region,
timed_region_with_gather, gather_solutions, max_over_ranks); only the solver object differs: the CPU
oracle stands in for BatchSolver here -- no GPU in this container. The gathered result must equal the
single-process answer in global problem order."""
import os
import socket
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _solve_shard(rank, world, per_rank, n, m, N):
    sys.path.insert(0, HERE)
    sys.path.insert(0, os.path.dirname(HERE))
    import rslqr_amd
    from rslqr_amd import sharding
    from support import Oracle, Problem
    orc = Oracle()
    seed0 = sharding.shard_seed0(rank, per_rank)
    out = []
    for p in range(per_rank):
        g = rslqr_amd.generate_synthetic(n, m, N, seed0 + p)
        prob = Problem(n, m, N, g["A"], g["B"], g["Q"], g["R"], g["q"], g["r"], g["d"], g["x0"])
        z, _, _, _ = orc.solve(prob, 1)
        out.append(z[: prob.nvars])
    return np.stack(out)


def _worker(rank, world, port, per_rank, n, m, N, outdir):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank),
                      WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sys.path.insert(0, os.path.dirname(HERE))
    from rslqr_amd import sharding
    assert sharding.env_rank() == (rank, rank, world)
    lo, hi = sharding.shard_range(rank, world, per_rank)
    assert (lo, hi) == (rank * per_rank, (rank + 1) * per_rank)

    class OracleShard:  # the surface of rslqr_amd.BatchSolver that the timed regions use
        calls = 0
        sol = None

        def solve_async(self):
            self.calls += 1
            self.sol = _solve_shard(rank, world, per_rank, n, m, N)

        def synchronize(self):
            return 0

        def solutions(self):
            return self.sol

    shard = OracleShard()
    elapsed = sharding.timed_region(shard, 2, 1, dist.barrier)
    assert shard.calls == 3 and elapsed > 0.0
    g_elapsed, allsol = sharding.timed_region_with_gather(
        shard, 1, dist.barrier, lambda: sharding.gather_solutions(shard.solutions()))
    assert shard.calls == 4 and g_elapsed > 0.0
    assert np.array_equal(allsol[rank * per_rank:(rank + 1) * per_rank], shard.solutions())
    mx = sharding.max_over_ranks([1.0 + rank, 5.0 - rank])
    if rank == 0:
        np.save(os.path.join(outdir, "all.npy"), allsol)
        np.save(os.path.join(outdir, "max.npy"), np.array(mx))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_sharding_matches_single_process(tmp_path):
    world, per_rank, n, m, N = 2, 3, 6, 3, 16
    for attempt in range(2):  # (the port is free when it is picked, not necessarily when rank 0 binds it: one retry)
        try:
            mp.spawn(_worker, args=(world, _free_port(), per_rank, n, m, N, str(tmp_path)), nprocs=world, join=True)
            break
        except Exception:  # noqa: BLE001
            if attempt:
                raise
    allsol = np.load(tmp_path / "all.npy")
    assert allsol.shape[0] == world * per_rank
    single = _solve_shard(0, 1, world * per_rank, n, m, N)  # global problems 0..5, seeds 1..6
    assert np.array_equal(allsol, single)
    assert np.array_equal(np.load(tmp_path / "max.npy"), [2.0, 5.0])


def _run_bench(args, env_extra, timeout):
    import json
    import subprocess
    env = dict(os.environ)
    env.pop("WORLD_SIZE", None)
    env.pop("RANK", None)
    env.pop("LOCAL_RANK", None)
    env.update(env_extra)
    root = os.path.dirname(HERE)
    proc = subprocess.run([sys.executable, os.path.join(root, "bench.py")] + args, env=env, cwd=root,
                          stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=timeout)
    lines = [json.loads(l) for l in proc.stdout.splitlines() if l.startswith("{")]
    return proc, lines


def test_bench_launches_its_own_ranks():
    """`python bench.py --gpus N` (N > 1, no launcher around it) starts N rank processes itself before anything
    touches a GPU, and a process group of another size than --gpus is an error. (Launch alone: the ranks stop
    before the solver is created -- NDLQR_BENCH_LAUNCH_ONLY -- so this runs without a GPU.)"""
    proc, lines = _run_bench(["--gpus", "2", "--steps", "3"], {"NDLQR_BENCH_LAUNCH_ONLY": "1"}, 300)
    assert proc.returncode == 0, proc.stderr[-2000:]
    assert sorted((l["rank"], l["world"], l["gpus"]) for l in lines) == [(0, 2, 2), (1, 2, 2)]
    # a launcher that made 2 ranks while the command line says --gpus 4: every rank exits 3
    proc, lines = _run_bench(["--gpus", "4"], {"NDLQR_BENCH_LAUNCH_ONLY": "1", "WORLD_SIZE": "2", "RANK": "0"}, 120)
    assert proc.returncode == 3


@pytest.mark.gpu
def test_bench_two_ranks_on_one_gpu():
    """bench.py's own N = 2 path end to end on the one GPU of the test box: two rank processes (launched by
    bench.py itself), both on device 0, gloo instead of RCCL (one GPU cannot host two RCCL ranks); the real
    BatchSolver, the shared timed regions, the gather leg."""
    proc, lines = _run_bench(["--gpus", "2", "--steps", "3", "--warmup", "1", "--batch", "64", "--no-modes",
                              "--spin-up-ms", "5", "--cpu-sample", "4", "--config4-batch", "8"],
                             {"NDLQR_BENCH_SAME_DEVICE": "1", "NDLQR_BENCH_BACKEND": "gloo",
                              "NDLQR_BENCH_FORCE_CONFIG4": "1"}, 900)
    assert proc.returncode == 0, proc.stderr[-3000:]
    assert len(lines) == 1, lines  # rank 0 alone prints
    res = lines[0]
    assert res["n_gpus"] == 2 and res["config"]["ranks_seen"] == 2
    assert res["scaling"] == "weak" and res["value"] > 0
    assert res["gather"]["every_rank_found_its_shard_intact"] is True
    assert res["config"]["kkt_residual_rel_max"] <= 1e-9 and res["config"]["cholesky_failures"] == 0
    # the N > 1 line is complete: roofline, the CPU column (rank 0), every rank's elapsed time, and BASELINE config 4
    # -- (12,4,1024) per GPU, gather included -- on every rank (at a reduced batch for this one-GPU rehearsal)
    assert 0.0 < res["roofline"]["frac"] <= 1.0 and res["roofline"]["kernel"]
    assert res["cpu_baseline"]["value"] > 0 and res["cpu_baseline"]["cores"] >= 1
    assert res["parity_rel_err_vs_cpu"] <= 1e-9
    assert len(res["elapsed_s_per_rank"]) == 2
    assert res["elapsed_minmax_s"][0] <= res["elapsed_minmax_s"][1] == max(res["elapsed_s_per_rank"])
    c4 = [v for k, v in res["configs"].items() if k.startswith("config4")]
    assert len(c4) == 1
    c4 = c4[0]
    assert c4["value"] > 0 and len(c4["elapsed_s_per_rank"]) == 2 and c4["is_baseline_config4"] is False
    assert c4["kkt_residual_rel_max"] <= 1e-9 and c4["cholesky_failures"] == 0
    assert c4["gather"]["every_rank_found_its_shard_intact"] is True and c4["gather"]["value_incl_gather"] > 0


@pytest.mark.gpu
def test_bench_rccl_path_one_rank():
    """The RCCL calls of the N > 1 path -- process-group init with a device id, barrier, MAX all_reduce on device
    tensors, all_gather_into_tensor of the packed solutions behind the solver's pack kernel -- executed for real
    on the one GPU of the test box: one rank under torch.distributed.run with NDLQR_BENCH_FORCE_DIST=1 (a one-rank
    group: the collectives are trivial, the API usage and stream ordering are not)."""
    import json
    import subprocess
    root = os.path.dirname(HERE)
    env = dict(os.environ, NDLQR_BENCH_FORCE_DIST="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr",
           "127.0.0.1", "--master-port", str(_free_port()), os.path.join(root, "bench.py"), "--gpus", "1", "--steps", "3",
           "--warmup", "1", "--batch", "64", "--no-cpu", "--no-modes", "--no-configs", "--no-transfers", "--spin-up-ms", "5"]
    proc = subprocess.run(cmd, env=env, cwd=root, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    assert proc.returncode == 0, proc.stderr[-3000:]
    lines = [json.loads(l) for l in proc.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    res = lines[0]
    assert res["n_gpus"] == 1 and res["config"]["ranks_seen"] == 1 and res["config"]["backend"] == "nccl"
    assert res["gather"]["every_rank_found_its_shard_intact"] is True
    assert "RCCL" in res["gather"]["collective"]
    assert res["config"]["kkt_residual_rel_max"] <= 1e-9

#!/usr/bin/env python3
"""bench.py -- headline benchmark: LQR solves/sec at (nx=12, nu=4, N=256, batch=1024 per GPU).

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One "step" = one ndlqr_SolveBatch (factor + substitute, the reference's ndlqr_Solve,
src/solve.c:38-190) over the rank's batch of independent synthetic problems, inputs already
resident in HBM. The batch axis is the sharding unit: every rank owns `--batch` problems
(weak scaling), there is no data-path collective; torch.distributed (RCCL) is used only for
the barriers and the max-over-ranks of the elapsed time.

Prints ONE JSON line on rank 0 (contract in the task description) with two extra objects:
  roofline     -- dominant kernel (the per-level kernel), algorithmic bytes of SURVEY.md 8(d)
                  model (B) for the levels it covers / its HIP-event time, vs 8 TB/s.
  cpu_baseline -- the reference's own ndlqr_Solve (oracle/_ref/libref.so, "reference") or the
                  plain-C oracle ("port") timed on this host on a bounded sample (N=1, rank 0).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md), ~6300 achievable


def model_b_bytes(n, m, N):
    """SURVEY.md 8(d) 'level-streaming' algorithmic bytes per solve: (leaves, [per level])."""
    K = int(np.log2(N))
    Fb = (2 * n + m) * n
    leaves = N * (4 * n * n + 3 * m * n + m * m + 5 * n + 3 * m)
    levels = []
    for l in range(K):
        L = 1 << (K - l - 1)
        P1 = L * (K - l) * (4 * n * n + 2 * m * n) + L * (n * n + n * m)
        P2 = 2 * L * n * n
        P3 = L * n * n + 2 * L * (K - l - 1) * n * n
        P4 = (N * Fb if l < K - 1 else 0) + L * (K - l - 1) * n * n + 2 * N * (K - l - 1) * Fb
        S = L * (2 * n * n + n * m + 8 * n + 2 * m) + N * (Fb + 5 * n + 2 * m)
        levels.append(8 * (P1 + P2 + P3 + P4 + S))
    return 8 * leaves, levels


def model_flops(n, m, N):
    """SURVEY.md 8(d) algorithmic flops per solve (dense reference schedule): leaves + per level
    P1 4n^2(n+m) per product, P2 n^3/3, P3 2n^3 per solve, P4 2*Fb*n per block update, plus the
    rhs sweep (w = 1)."""
    K = int(np.log2(N))
    Fb = (2 * n + m) * n
    fl = N * (n ** 3 / 3 + m ** 3 / 3 + 4 * n ** 3 + 2 * m * m * n)
    for l in range(K):
        L = 1 << (K - l - 1)
        fl += L * (K - l) * 4 * n * n * (n + m) + L * n ** 3 / 3 + L * (K - l - 1) * 2 * n ** 3
        fl += N * (K - l - 1) * 2 * Fb * n
        fl += L * (4 * n * (n + m) + 2 * n * n) + N * 2 * Fb
    return fl


def live_bytes(n, m, N):
    """Bytes the kernels of this build actually have to move per solve (DESIGN.md "live
    columns"): inputs + rhs once, per level read E + read/write one outer column + write the
    other + rhs read/write, plus the separator blocks."""
    K = int(np.log2(N))
    Fb = (2 * n + m) * n
    zb = 2 * n + m
    leaves = N * (n * (n + m) + (n + m) + zb) + N * (2 * Fb + zb)
    total = leaves
    for l in range(K):
        L = 1 << (K - l - 1)
        cols = 0 if l == K - 1 else 2  # outer columns alive (upper bound; ends have one)
        sep = L * (n * (n + m) + 2 * Fb + 2 * zb + (1 + cols) * n * n + n)
        schur = N * (Fb + cols * Fb + (cols - 1 if cols else 0) * Fb + 2 * zb)
        total += sep + schur
    return 8 * total


def host_cores():
    """CPU cores this process may really use: min(affinity mask, cgroup cpu quota). On the GPU
    boxes the mask shows every host CPU while the cgroup grants a share (16 per GPU)."""
    cores = len(os.sched_getaffinity(0))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            cores = min(cores, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    env = os.environ.get("NDLQR_BENCH_CORES")
    return int(env) if env else cores


def log(msg):
    print("[bench] " + msg, file=sys.stderr, flush=True)


def cpu_baseline(n, m, N, probs, gpu_solutions):
    """Times the CPU checker on a bounded sample; returns (dict, parity_rel_err)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import support  # the oracle bindings: used here ONLY as checker / baseline
    cores = host_cores()
    flat = [np.ascontiguousarray(np.stack([p[k] for p in probs])) for k in
            ("A", "B", "Q", "R", "q", "r", "d", "x0")]
    count = len(probs)
    import ctypes as C
    args = [a.ctypes.data_as(support.dp) for a in flat]
    # parity of the GPU result against the checker on the same sample
    orc = support.Oracle()
    worst = 0.0
    for i, p in enumerate(probs):
        prob = support.Problem(n, m, N, p["A"], p["B"], p["Q"], p["R"], p["q"], p["r"], p["d"], p["x0"])
        z, _, _, _ = orc.solve(prob, 1)
        ref = z[: prob.nvars]
        worst = max(worst, float(np.linalg.norm(gpu_solutions[i] - ref) / np.linalg.norm(ref)))
    out = {"cores": cores}
    if support.have_reference():
        ref = support.Reference()
        # (i) reference semantics: one solve at a time, all cores inside the solve
        ref.L.ref_bench(n, m, N, min(count, 2), 1, *args, cores)  # warm-up
        ms_i = ref.L.ref_bench(n, m, N, count, 1, *args, cores)
        rate_i = count / (ms_i * 1e-3)
        # (ii) throughput: one thread per solve, all cores busy with different problems
        ref.L.ref_bench_throughput.restype = C.c_double
        ref.L.ref_bench_throughput.argtypes = [C.c_int] * 5 + [support.dp] * 8 + [C.c_int]
        ref.L.ref_bench_throughput(n, m, N, min(count, cores), 1, *args, cores)
        reps = 3
        ms_ii = ref.L.ref_bench_throughput(n, m, N, count, reps, *args, cores)
        rate_ii = count * reps / (ms_ii * 1e-3)
        out.update(kind="reference", value=max(rate_i, rate_ii), unit="solves/s",
                   sample="%d problems of (%d,%d,%d): reference ndlqr_Solve from oracle/_ref; "
                          "(i) 1 solve at a time x %d threads = %.1f solves/s, (ii) %d solves in "
                          "parallel x 1 thread = %.1f solves/s; value = the better"
                          % (count, n, m, N, cores, rate_i, cores, rate_ii))
    else:
        ms = C.c_double(0)
        orc.L.oracle_bench(n, m, N, min(count, cores), 1, *args, cores, 1, C.byref(ms))
        reps = 4
        orc.L.oracle_bench(n, m, N, count, reps, *args, cores, 1, C.byref(ms))
        rate_ii = count * reps / (ms.value * 1e-3)
        orc.L.oracle_bench(n, m, N, count, 1, *args, cores, 0, C.byref(ms))
        rate_i = count / (ms.value * 1e-3)
        out.update(kind="port", value=max(rate_i, rate_ii), unit="solves/s",
                   sample="%d problems of (%d,%d,%d): plain-C oracle (oracle/ndlqr_oracle.c); (i) 1 "
                          "solve at a time x %d threads = %.1f solves/s, (ii) %d solves in parallel x "
                          "1 thread = %.1f solves/s; value = the better"
                          % (count, n, m, N, cores, rate_i, cores, rate_ii))
    return out, worst


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--nx", type=int, default=12)
    ap.add_argument("--nu", type=int, default=4)
    ap.add_argument("--horizon", type=int, default=256)
    ap.add_argument("--batch", type=int, default=1024, help="problems per GPU")
    ap.add_argument("--flags", type=int, default=0, help="NDLQR_FLAG_* bits (1 strict, 2 generic)")
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    ap.add_argument("--cpu-sample", type=int, default=0, help="problems in the CPU sample (0 = auto)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    from rslqr_amd import sharding
    rank, local_rank, world = sharding.env_rank()
    if not torch.cuda.is_available():
        print("bench.py needs a GPU (no CPU fallback for the product path)", file=sys.stderr)
        sys.exit(2)
    # Rehearsal hooks (not used by the driver): NDLQR_BENCH_SAME_DEVICE=1 puts every rank on GPU 0
    # and NDLQR_BENCH_BACKEND=gloo swaps RCCL for gloo, so the N>1 control flow can be exercised
    # on a one-GPU box.
    if os.environ.get("NDLQR_BENCH_SAME_DEVICE"):
        local_rank = 0
    backend = os.environ.get("NDLQR_BENCH_BACKEND", "nccl")
    torch.cuda.set_device(local_rank)
    distributed = world > 1
    if distributed:
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)

    import rslqr_amd
    n, m, N, batch = args.nx, args.nu, args.horizon, args.batch
    bs = rslqr_amd.BatchSolver(n, m, N, batch, device=local_rank, flags=args.flags)
    seed0 = sharding.shard_seed0(rank, batch)  # global problem g has seed 1 + g
    log("rank %d: generating + uploading %d synthetic problems" % (rank, batch))
    bs.initialize_synthetic(seed0)
    log("rank %d: warm-up" % rank)

    def barrier():
        if distributed:
            dist.barrier()
        torch.cuda.synchronize()

    # Timed region: the product path as shipped (launch sequence replayed as a hipGraph).
    for _ in range(args.warmup):
        bs.solve_async()
    bs.synchronize()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        bs.solve_async()
    bs.synchronize()
    barrier()
    elapsed = time.perf_counter() - t0
    log("rank %d: %d steps in %.3f s" % (rank, args.steps, elapsed))
    fails = bs.cholesky_failures()

    # Per-kernel durations for the roofline object: the SAME K steps once more with a HIP-event
    # pair around every launch on the launch stream. (Events force eager launches, so they cannot
    # sit inside the graph-replayed region above; the kernels and their inputs are identical.)
    bs.set_flags(args.flags | rslqr_amd.FLAG_PROFILE)
    bs.solve()
    bs.profile_reset()
    tp0 = time.perf_counter()
    for _ in range(args.steps):
        bs.solve_async()
    bs.synchronize()
    elapsed_profiled = time.perf_counter() - tp0
    prof = bs.profile()
    bs.set_flags(args.flags)

    # per-solve device times (HIP events around the replayed launch sequence, one solve at a time):
    # median and minimum next to the mean of the timed region
    per_solve = []
    for _ in range(min(args.steps, 50)):
        bs.solve()
        per_solve.append(bs.solve_ms())
    per_solve.sort()
    dev_median, dev_min = per_solve[len(per_solve) // 2], per_solve[0]

    # every problem of the shard, checked on the device against its raw data (outside the timed
    # region): worst ||K z - b|| / max(1, ||b||) -- SURVEY.md 8(d) "KKT residual of every problem"
    kres, kbn = bs.kkt_residuals()
    kkt_worst = float((kres / [max(1.0, v) for v in kbn]).max())

    elapsed_max, fails_max, kkt_max = sharding.max_over_ranks(
        [elapsed, float(fails), kkt_worst], device="cuda" if backend == "nccl" else "cpu")
    fails_max = int(fails_max)

    if rank == 0:
        total_solves = batch * world * args.steps
        value = total_solves / elapsed_max
        leaf_b, level_b = model_b_bytes(n, m, N)
        # Dominant kernel = the slot with the largest HIP-event time. Its algorithmic bytes are the
        # SURVEY 8(d) model-(B) bytes of exactly the phases it covers:
        #   bottom          leaf phase + levels 0..JB-1            (1 launch per step)
        #   apply           Schur/solution sweep of levels J..K-1  (1 launch per step)
        #   separator+schur one level each (generic / level-by-level path)
        K = len(level_b)
        used = {k: v for k, v in prof.items() if v[1] > 0}
        dom = max(used, key=lambda k: used[k][0])
        JB = int(os.environ.get("NDLQR_BOTTOM_LEVELS", "2"))
        if dom == "bottom":
            covered = leaf_b + sum(level_b[:JB])
            dom_ms, dom_launches = used["bottom"]
        elif dom == "apply":
            covered = sum(level_b[JB:])
            dom_ms, dom_launches = used["apply"]
        elif dom == "upper":
            # separators + boundary Schur of levels JB..K-1 in one launch; model (B) charges those
            # levels' P1-P3 bytes to it and the Schur/solution sweep to apply -- reported together
            covered = sum(level_b[JB:])
            dom_ms, dom_launches = used["upper"]
        else:
            dom_ms = sum(used[k][0] for k in ("separator", "schur") if k in used)
            dom_launches = used.get("schur", used.get("separator"))[1]
            covered = sum(level_b) / K
            dom = "separator+schur"
        avg_ms = dom_ms / max(dom_launches, 1)
        bytes_per_launch = covered * batch
        achieved = bytes_per_launch / (avg_ms * 1e-3) / 1e9
        # measured HBM bytes per launch of that kernel: committed PMC summary of the same config
        # (rocprofv3 cannot run inside this process; profiles/r01_traffic.json says how it was taken)
        traffic, issue = None, None
        try:
            tj = json.load(open(os.path.join(ROOT, "profiles", "r01_traffic.json")))
            if (n, m, N, batch, args.flags) == (12, 4, 256, 1024, 0) and dom in tj["kernels"]:
                traffic = tj["kernels"][dom]["hbm_bytes_per_launch"]
                issue = tj.get("issue", {}).get(dom)
                if issue and "executed_flops_per_wave" in issue and dom == "bottom":
                    # executed (not algorithmic) fp64 rate of the dominant kernel: flops per wavefront
                    # from the ISA x wavefronts of this launch / its measured duration
                    issue = dict(issue)
                    waves = (N // 4) * batch
                    issue["executed_tflops"] = issue["executed_flops_per_wave"] * waves / (avg_ms * 1e-3) / 1e12
                    issue["executed_frac_of_fp64_peak"] = issue["executed_tflops"] / 78.6
        except (OSError, ValueError, KeyError):
            pass
        whole_solve_gbs = (leaf_b + sum(level_b)) * value / world / 1e9
        result = {
            "metric": "LQR solves/sec (nx=%d,nu=%d,N=%d,batch=%d per GPU)" % (n, m, N, batch),
            "value": value, "unit": "solves/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": elapsed_max / args.steps * 1e3,
            "ms_per_solve": 1e3 / value,
            "device_ms_per_step": {"median": dev_median, "min": dev_min, "reps": len(per_solve),
                                   "note": "rank 0, HIP events per solve, one solve in flight"},
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64",
            "data": "synthetic (seeded splitmix64 family of SURVEY.md 8d, time-varying A,B,Q,R)",
            "config": {"workload": "nx=%d nu=%d N=%d batch=%d per GPU, fp64, factor+solve per step"
                                   % (n, m, N, batch),
                       "parallelism": "batch-sharded x%d, no data-path collective" % world,
                       "flags": args.flags, "cholesky_failures": fails_max,
                       "kkt_residual_rel_max": kkt_max},
            "roofline": {"bound": "hbm", "kernel": dom, "achieved": achieved, "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "avg_launch_ms": avg_ms, "launches": dom_launches,
                         "algorithmic_bytes_per_launch": bytes_per_launch,
                         "whole_solve_model_b_gbs_per_gpu": whole_solve_gbs,
                         "live_column_bytes_per_solve": live_bytes(n, m, N),
                         # the kernels are fp64 VALU-issue bound after the traffic reduction: the
                         # compute roof next to the (contractual) HBM one. Algorithmic flops of
                         # the dense reference schedule; structural zeros are skipped at run time.
                         "fp64": {"algorithmic_flops_per_solve": model_flops(n, m, N),
                                  "achieved_tflops": model_flops(n, m, N) * value / world / 1e12,
                                  "peak_tflops": 78.6,
                                  "frac": model_flops(n, m, N) * value / world / 1e12 / 78.6},
                         "model_b_bytes_per_solve": leaf_b + sum(level_b),
                         # what actually bounds the dominant kernel (SQ counters of the committed
                         # profile, not measured in this run): fp64 vector-ALU + matrix issue time
                         "issue_bound_profile": issue},
            "kernel_ms": {k: {"total_ms": v[0], "launches": v[1]} for k, v in prof.items() if v[1]},
            "kernel_ms_note": "second pass of the same %d steps with per-launch HIP events (eager "
                              "launches): %.3f ms/step vs %.3f ms/step in the timed, graph-replayed region"
                              % (args.steps, elapsed_profiled / args.steps * 1e3, elapsed_max / args.steps * 1e3),
        }
        if world == 1 and not args.no_cpu:
            cores = host_cores()
            log("cpu_baseline leg on %d cores" % cores)
            count = args.cpu_sample or max(8, min(batch, 8 * cores))  # ~10 s of CPU work
            probs = [rslqr_amd.generate_synthetic(n, m, N, seed0 + p) for p in range(count)]
            gpu_sol = [bs.solution(p) for p in range(count)]
            base, worst = cpu_baseline(n, m, N, probs, gpu_sol)
            result["cpu_baseline"] = base
            result["parity_rel_err_vs_cpu"] = worst
            result["speedup_vs_cpu"] = value / base["value"]
        print(json.dumps(result), flush=True)

    bs.close()
    if distributed:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

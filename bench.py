#!/usr/bin/env python3
"""bench.py -- headline benchmark: LQR solves/sec at (nx=12, nu=4, N=256, batch=1024 per GPU).

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python bench.py --nx 64 --nu 16 --horizon 512 --batch 256        # BASELINE config 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One "step" = one ndlqr_SolveBatch (factor + substitute, the reference's ndlqr_Solve,
src/solve.c:38-190) over the rank's batch of independent synthetic problems, inputs already
resident in HBM. The batch axis is the sharding unit: every rank owns `--batch` problems (weak
scaling), there is no data-path collective; torch.distributed (RCCL) carries the barriers, the
max-over-ranks of the elapsed time and -- outside `value` -- the optional gather of the solutions.

Prints ONE JSON line on rank 0 (contract in the task description) with:
  roofline     -- dominant kernel of the step: algorithmic bytes and useful flops of the IMPLEMENTED
                  schedule (rslqr_amd/roofline.py) / its HIP-event launch time
                  measured in this run, against 8 TB/s and 78.6 TFLOP/s fp64; the larger fraction
                  labels the bound. `kernels` holds the same for every kernel kind, `step` for the whole
                  step, `traffic` the PMC-measured HBM bytes per launch of the committed profile of
                  exactly this source tree (null when the tree differs).
  cpu_baseline -- the reference's own ndlqr_Solve (oracle/_ref/libref.so, "reference") or the
                  plain-C oracle ("port") timed on this host on a bounded sample (N=1, rank 0).
  pipeline     -- depth of the solve pipeline behind `value` and the same steps strictly stream-ordered.
  transfers    -- H2D of the packed inputs and D2H of the solutions, timed separately (never in `value`).
  end_to_end   -- the MPC step of a drop-in caller: right-hand side up, factor + solve, solutions down, two in flight.
  spin_up      -- untimed full solves before the W warm-up steps (one-time work of the first solves, clock ramp).
  modes        -- N=1: strict mode, KEEP_FACT and the rhs-only re-solve on the same workload.
  configs      -- N=1: the other single-GPU configurations of BASELINE.json, each timed the same way on a bounded
                  number of steps: lqr_prob_256.json x 1 (latency, error against the fixture's `soln`),
                  (12,4,1024) x 512 (one GPU's shard of config 4), (64,16,512) x 256 (config 5).
  gather       -- N>1: throughput including the all_gather of every shard's solutions.

`python bench.py --gpus N` with N > 1 and no WORLD_SIZE in the environment starts its own N ranks: this
process -- before it imports torch or touches a GPU in any way -- runs `python -m torch.distributed.run
--nnodes=1 --nproc-per-node N ... bench.py <same arguments>` as a child, relays rank 0's JSON line and exits
with the child's status. A process group whose size differs from --gpus is an error (exit 3), not a warning.
"""
import argparse
import hashlib
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from rslqr_amd import roofline as rf  # noqa: E402


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


def csrc_sha():
    """Content hash of the kernel sources: ties a committed PMC summary to the tree it was taken on
    (the GPU box has no .git)."""
    h = hashlib.sha256()
    d = os.path.join(ROOT, "rslqr_amd", "csrc")
    for name in sorted(os.listdir(d)):
        if name.endswith((".hpp", ".hip", ".c", ".def", ".h")):
            h.update(name.encode())
            h.update(open(os.path.join(d, name), "rb").read())
    return h.hexdigest()[:16]


def committed_traffic(n, m, N, batch, flags):
    """{slot: HBM bytes per launch} from the newest profiles/*_traffic.json taken on exactly this
    source tree and workload, else ({}, why)."""
    pdir = os.path.join(ROOT, "profiles")
    sha = csrc_sha()
    best = None
    for name in sorted(os.listdir(pdir)) if os.path.isdir(pdir) else []:
        if not name.endswith("_traffic.json"):
            continue
        try:
            tj = json.load(open(os.path.join(pdir, name)))
        except (OSError, ValueError):
            continue
        if tj.get("workload") == [n, m, N, batch, flags] and (best is None or tj.get("csrc_sha") == sha or
                                                              best[1].get("csrc_sha") != sha):
            best = (name, tj)  # the newest one taken on this source tree, else the newest one
    if best is None:
        return {}, "no committed PMC summary for this workload"
    name, tj = best
    if tj.get("csrc_sha") != sha:
        return {}, "%s was taken on csrc %s, this tree is %s: not printed" % (name, tj.get("csrc_sha"), sha)
    return {k: v["hbm_bytes_per_launch"] for k, v in tj["kernels"].items()}, "profiles/" + name


def committed_issue(n, m, N, batch, flags):
    """{slot: entry of profiles/*_issue.json} (instructions per wavefront, fp64-pipe time per solve: tools/make_issue.py)
    taken on exactly this source tree and workload, else ({}, why)."""
    pdir = os.path.join(ROOT, "profiles")
    sha = csrc_sha()
    for name in sorted(os.listdir(pdir), reverse=True) if os.path.isdir(pdir) else []:
        if not name.endswith("_issue.json"):
            continue
        try:
            tj = json.load(open(os.path.join(pdir, name)))
        except (OSError, ValueError):
            continue
        if tj.get("workload") == [n, m, N, batch, flags] and tj.get("csrc_sha") == sha:
            return tj["kernels"], "profiles/" + name
    return {}, "no committed SQ-counter summary of this tree for this workload"


def cpu_baseline(n, m, N, probs, gpu_solutions):
    """Times the CPU checker on a bounded sample; returns (dict, parity_rel_err)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import support  # the oracle bindings: used here ONLY as checker / baseline
    import ctypes as C
    cores = host_cores()
    flat = [np.ascontiguousarray(np.stack([p[k] for p in probs])) for k in
            ("A", "B", "Q", "R", "q", "r", "d", "x0")]
    count = len(probs)
    args = [a.ctypes.data_as(support.dp) for a in flat]
    big = rf.model_b_flops(n, m, N) > 2e9  # (64,16,512): seconds per reference solve
    # parity of the GPU result against the checker on the same sample
    orc = support.Oracle()
    worst = 0.0
    for i, p in enumerate(probs[: 1 if big else count]):
        prob = support.Problem(n, m, N, p["A"], p["B"], p["Q"], p["R"], p["q"], p["r"], p["d"], p["x0"])
        z, _, _, _ = orc.solve(prob, cores if big else 1)
        ref = z[: prob.nvars]
        worst = max(worst, float(np.linalg.norm(gpu_solutions[i] - ref) / np.linalg.norm(ref)))
    out = {"cores": cores}
    if support.have_reference():
        ref = support.Reference()
        ref.L.ref_bench_throughput.restype = C.c_double
        ref.L.ref_bench_throughput.argtypes = [C.c_int] * 5 + [support.dp] * 8 + [C.c_int]
        # (i) reference semantics: one solve at a time, all cores inside the solve
        ref.L.ref_bench(n, m, N, 1 if big else min(count, 2), 1, *args, cores)  # warm-up
        ms_i = ref.L.ref_bench(n, m, N, count, 1, *args, cores)
        rate_i = count / (ms_i * 1e-3)
        if big:
            out.update(kind="reference", value=rate_i, unit="solves/s",
                       sample="%d problems of (%d,%d,%d): reference ndlqr_Solve from oracle/_ref, 1 solve at "
                              "a time x %d threads (the one-thread-per-solve mode would take minutes here)"
                              % (count, n, m, N, cores))
            return out, worst
        # (ii) throughput: one thread per solve, all cores busy with different problems
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
        reps = 1 if big else 4
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


def riccati_column():
    """the reference's serial Riccati comparison solver (src/riccati_solve.c:7-150, compiled into
    oracle/_ref) next to the reference's ndlqr_Solve, on the two JSON fixtures of the reference (its own comparison,
    test/sample_problem_test.c:127-177; off-fixture it diverges on long random horizons -- SURVEY.md App. C).
    One thread: Riccati is serial."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import support
    if not support.have_reference():
        return None
    ref = support.Reference()
    out = {"kind": "reference", "cores": 1, "note": "reference ndlqr_SolveRiccati vs reference ndlqr_Solve (1 thread), "
           "JSON fixtures only"}
    for fname in ("lqr_prob.json", "lqr_prob_256.json"):
        prob, soln = support.load_json_problem(os.path.join(support.GOLDEN, fname))
        x, ms = ref.riccati(prob, reps=20)
        flat = [np.ascontiguousarray(a[None]) for a in prob.arrays()]
        args = [a.ctypes.data_as(support.dp) for a in flat]
        ref.L.ref_bench(prob.n, prob.m, prob.N, 1, 2, *args, 1)
        nd_ms = ref.L.ref_bench(prob.n, prob.m, prob.N, 1, 10, *args, 1) / 10
        out["%s (%d,%d,%d)" % (fname, prob.n, prob.m, prob.N)] = {
            "riccati_ms_per_solve": ms, "riccati_solves_per_s": 1e3 / ms,
            "riccati_err_vs_fixture_soln_l2": float(np.linalg.norm(x - soln)),
            "rslqr_1thread_ms_per_solve": nd_ms, "rslqr_1thread_solves_per_s": 1e3 / nd_ms}
    return out


def transfer_legs(rslqr_amd, bs, n, m, N, batch, seed0, steps):
    """`transfers`: H2D of the packed inputs and D2H of the solutions, pageable and pinned host memory, whole shard,
    each timed on its own. `end_to_end`: the MPC step a drop-in caller runs -- new right-hand side up, factor + solve,
    solutions down (ndlqr_BatchStepAsync), two steps in flight on the two buffer sets of the pipeline, pinned host
    arrays -- timed over `steps` steps; never part of `value`."""
    w, rows = n + m, 2 * n + m
    in_bytes = 8 * batch * N * (n * w + w + rows)
    out = {"h2d_bytes": in_bytes, "d2h_bytes": 8 * batch * bs.nvars}
    # D2H of the solutions
    t0 = time.perf_counter()
    sol = bs.solutions()
    out["d2h_ms_pageable_first_touch"] = (time.perf_counter() - t0) * 1e3
    t0 = time.perf_counter()
    bs.solutions(out=sol)
    out["d2h_ms_pageable"] = (time.perf_counter() - t0) * 1e3
    pin = rslqr_amd.pinned_empty((batch, bs.nvars))
    bs.solutions(out=pin)
    t0 = time.perf_counter()
    bs.solutions(out=pin)
    out["d2h_ms"] = (time.perf_counter() - t0) * 1e3
    out["d2h_gbs"] = out["d2h_bytes"] / out["d2h_ms"] / 1e6
    out["d2h_equal"] = bool(np.array_equal(pin, sol))
    # end to end: the same problems' own right-hand sides, so the result must equal the resident solution
    q, r, d, x0 = (rslqr_amd.pinned_empty(s) for s in ((batch, N, n), (batch, N, m), (batch, N, n), (batch, n)))
    for p in range(batch):
        g = rslqr_amd.generate_synthetic(n, m, N, seed0 + p)
        q[p], r[p], d[p], x0[p] = g["q"], g["r"], g["d"], g["x0"]
    outs = [rslqr_amd.pinned_empty((batch, bs.nvars)) for _ in range(2)]

    def run(k, full, dst=None, x=None):
        dst = dst or outs
        x = x0 if x is None else x
        for i in range(k):
            err = bs.step_async(q, r, d, x, dst[i & 1]) if full else bs.step_async(None, None, None, x, dst[i & 1])
            if err != 0:
                raise RuntimeError("ndlqr_BatchStepAsync failed")
            if i >= 1:
                bs.synchronize_previous()
        bs.synchronize()

    end_to_end = {"steps": steps, "d2h_bytes_per_step": 8 * batch * bs.nvars,
                  "note": "ndlqr_BatchStepAsync, pinned host arrays, two steps in flight, rank 0; not part of `value`. "
                          "full_rhs: q, r, d, x0 read over the host link by the pack kernel, factor + solve, pack kernel, "
                          "solutions down; x0_only: the same with x0 alone replaced (the usual MPC iteration); "
                          "x0_only_u0: x0 up, and of the solutions only u of knot 0 down (ndlqr_BatchSetStepSelection: what "
                          "an MPC loop applies; [batch][m] doubles); x0_only_u0_computed_alone: the same with "
                          "NDLQR_SOLN_ONLY -- the step computes nothing but the eight knots around knot 0 in its last launch; "
                          "device_resident_*: x0 and u0 in device memory (ndlqr_DeviceAlloc), no transfer; "
                          "device_resident_new_problem_*: A, B, Q, R, q, r, d, x0 packed from device memory every step "
                          "(ndlqr_InitializeBatchFlatDevice + ndlqr_SolveBatchSlicesAsync)"}
    for name, full in (("full_rhs", True), ("x0_only", False)):
        run(4, full)
        t0 = time.perf_counter()
        run(steps, full)
        e2e = (time.perf_counter() - t0) / steps
        end_to_end[name] = {"ms_per_step": e2e * 1e3, "solves_per_s": batch / e2e,
                            "h2d_bytes_per_step": 8 * batch * ((N * rows if full else 0) + n),
                            "equals_resident_solution": bool(np.array_equal(outs[(steps - 1) & 1], sol))}
    # what an MPC loop consumes: u of knot 0 (the full vector stays resident: ndlqr_CopyBatchSolutionSlices / Solutions)
    bs.set_step_selection(0, 1, rslqr_amd.SOLN_INPUT)
    u0 = [rslqr_amd.pinned_empty((batch, 1, m)) for _ in range(2)]
    run(4, False, u0)
    t0 = time.perf_counter()
    run(steps, False, u0)
    e2e = (time.perf_counter() - t0) / steps
    end_to_end["x0_only_u0"] = {"ms_per_step": e2e * 1e3, "solves_per_s": batch / e2e, "h2d_bytes_per_step": 8 * batch * n,
                                "d2h_bytes_per_step": 8 * batch * m,
                                "equals_resident_solution": bool(np.array_equal(
                                    u0[(steps - 1) & 1][:, 0, :], sol[:, 2 * n:2 * n + m]))}
    # ... and computes nothing else (NDLQR_SOLN_ONLY: the last launch of the back-substitution runs one workgroup per problem)
    bs.set_step_selection(0, 1, rslqr_amd.SOLN_INPUT | rslqr_amd.SOLN_ONLY)
    run(4, False, u0)
    t0 = time.perf_counter()
    run(steps, False, u0)
    e2e = (time.perf_counter() - t0) / steps
    end_to_end["x0_only_u0_computed_alone"] = {
        "ms_per_step": e2e * 1e3, "solves_per_s": batch / e2e, "h2d_bytes_per_step": 8 * batch * n,
        "d2h_bytes_per_step": 8 * batch * m, "schedule": bs.schedule(),
        "equals_resident_solution": bool(np.array_equal(u0[(steps - 1) & 1][:, 0, :], sol[:, 2 * n:2 * n + m]))}
    # ... with x0 and u0 in device memory (the loop around the solver lives on the GPU: nothing crosses the host link)
    dx0 = rslqr_amd.DeviceArray((batch, n)).set(x0)
    du0 = [rslqr_amd.DeviceArray((batch, 1, m)) for _ in range(2)]
    run(4, False, du0, dx0)
    t0 = time.perf_counter()
    run(steps, False, du0, dx0)
    e2e = (time.perf_counter() - t0) / steps
    end_to_end["device_resident_x0_u0_computed_alone"] = {
        "ms_per_step": e2e * 1e3, "solves_per_s": batch / e2e, "h2d_bytes_per_step": 0, "d2h_bytes_per_step": 0,
        "schedule": bs.schedule(),
        "equals_resident_solution": bool(np.array_equal(du0[(steps - 1) & 1].get()[:, 0, :], sol[:, 2 * n:2 * n + m]))}
    # ... and with the WHOLE right-hand side replaced from device memory (an outer loop -- ADMM, SQP -- that lives on the GPU)
    dq, dr, dd = (rslqr_amd.DeviceArray(a.shape).set(a) for a in (q, r, d))

    def run_dev(k, dst):
        for i in range(k):
            if bs.step_async(dq, dr, dd, dx0, dst[i & 1]) != 0:
                raise RuntimeError("ndlqr_BatchStepAsync failed")
            if i >= 1:
                bs.synchronize_previous()
        bs.synchronize()
    run_dev(4, du0)
    t0 = time.perf_counter()
    run_dev(steps, du0)
    e2e = (time.perf_counter() - t0) / steps
    end_to_end["device_resident_full_rhs_u0_computed_alone"] = {
        "ms_per_step": e2e * 1e3, "solves_per_s": batch / e2e, "h2d_bytes_per_step": 0, "d2h_bytes_per_step": 0,
        "equals_resident_solution": bool(np.array_equal(du0[(steps - 1) & 1].get()[:, 0, :], sol[:, 2 * n:2 * n + m]))}
    bs.set_step_selection()
    dsol = [rslqr_amd.DeviceArray((batch, bs.nvars)) for _ in range(2)]
    run_dev(4, dsol)
    t0 = time.perf_counter()
    run_dev(steps, dsol)
    e2e = (time.perf_counter() - t0) / steps
    end_to_end["device_resident_full_rhs"] = {
        "ms_per_step": e2e * 1e3, "solves_per_s": batch / e2e, "h2d_bytes_per_step": 0, "d2h_bytes_per_step": 0,
        "equals_resident_solution": bool(np.array_equal(dsol[(steps - 1) & 1].get(), sol))}
    del dq, dr, dd, dsol
    # ... and with A, B, Q, R replaced as well (a loop that re-linearises on the GPU): the pack kernel takes the eight flat
    # arrays from device memory, ndlqr_SolveBatchSlicesAsync factors, solves and delivers u of knot 0 alone. The inputs are
    # shared by the two buffer sets, so every iteration waits for the one before.
    gens = [rslqr_amd.generate_synthetic(n, m, N, seed0 + p) for p in range(batch)]
    dall = [rslqr_amd.DeviceArray((batch,) + gens[0][k].shape).set(np.stack([g[k] for g in gens]))
            for k in ("A", "B", "Q", "R", "q", "r", "d", "x0")]
    del gens

    def run_new(k):
        for i in range(k):
            bs.initialize_flat_device(*[a.ptr for a in dall])
            if bs.solve_slices_async(0, 1, rslqr_amd.SOLN_INPUT, du0[i & 1]) != 0:
                raise RuntimeError("ndlqr_SolveBatchSlicesAsync failed")
        bs.synchronize()
    run_new(3)
    t0 = time.perf_counter()
    run_new(steps)
    e2e = (time.perf_counter() - t0) / steps
    end_to_end["device_resident_new_problem_u0_computed_alone"] = {
        "ms_per_step": e2e * 1e3, "solves_per_s": batch / e2e, "h2d_bytes_per_step": 0, "d2h_bytes_per_step": 0,
        "equals_resident_solution": bool(np.array_equal(du0[(steps - 1) & 1].get()[:, 0, :], sol[:, 2 * n:2 * n + m]))}
    del dall
    if bs.solve() != 0:  # (the solver holds the whole solution vector again)
        raise RuntimeError("ndlqr_SolveBatch failed")
    # the same loop with the factorisation kept (NDLQR_FLAG_KEEP_RECORDS): a step never changes A, B, Q, R, so every step
    # after the first is the right-hand-side re-solve on the compact records (stream-ordered: one step in flight)
    lti = rslqr_amd.BatchSolver(n, m, N, batch, flags=rslqr_amd.FLAG_KEEP_RECORDS)
    try:
        lti.initialize_synthetic(seed0)
        lti.set_step_selection(0, 1, rslqr_amd.SOLN_INPUT)

        def run_lti(k):
            for i in range(k):
                if lti.step_async(None, None, None, x0, u0[i & 1]) != 0:
                    raise RuntimeError("ndlqr_BatchStepAsync failed")
                if i >= 1:
                    lti.synchronize_previous()
            lti.synchronize()
        run_lti(4)
        t0 = time.perf_counter()
        run_lti(steps)
        e2e = (time.perf_counter() - t0) / steps
        end_to_end["x0_only_u0_records_kept"] = {
            "ms_per_step": e2e * 1e3, "solves_per_s": batch / e2e, "h2d_bytes_per_step": 8 * batch * n,
            "d2h_bytes_per_step": 8 * batch * m, "schedule": lti.schedule(),
            "equals_resident_solution_to": float(np.abs(u0[(steps - 1) & 1][:, 0, :] - sol[:, 2 * n:2 * n + m]).max())}
        lti.set_step_selection(0, 1, rslqr_amd.SOLN_INPUT | rslqr_amd.SOLN_ONLY)
        run_lti(4)
        t0 = time.perf_counter()
        run_lti(steps)
        e2e = (time.perf_counter() - t0) / steps
        end_to_end["x0_only_u0_computed_alone_records_kept"] = {
            "ms_per_step": e2e * 1e3, "solves_per_s": batch / e2e, "h2d_bytes_per_step": 8 * batch * n,
            "d2h_bytes_per_step": 8 * batch * m, "schedule": lti.schedule(),
            "equals_resident_solution_to": float(np.abs(u0[(steps - 1) & 1][:, 0, :] - sol[:, 2 * n:2 * n + m]).max())}
    finally:
        lti.close()
    end_to_end["ms_per_step"] = end_to_end["full_rhs"]["ms_per_step"]
    end_to_end["solves_per_s"] = end_to_end["full_rhs"]["solves_per_s"]
    # H2D of the packed inputs (a fresh solver: the upload replaces the inputs)
    h2d = {}
    if in_bytes <= (8 << 30):
        tmp = rslqr_amd.BatchSolver(n, m, N, batch)
        try:
            for kind in ("pageable", "pinned"):
                mk = (lambda s: np.ones(s)) if kind == "pageable" else (lambda s: rslqr_amd.pinned_empty(s))
                hAB, hQR, hrhs = mk((batch, N, n * w)), mk((batch, N, w)), mk((batch, N, rows))
                if kind == "pinned":
                    hAB[...], hQR[...], hrhs[...] = 1.0, 1.0, 1.0
                tmp.upload_packed(hAB, hQR, hrhs)
                t0 = time.perf_counter()
                tmp.upload_packed(hAB, hQR, hrhs)
                h2d[kind] = (time.perf_counter() - t0) * 1e3
                del hAB, hQR, hrhs
        finally:
            tmp.close()
        out["h2d_ms_pageable"] = h2d["pageable"]
        out["h2d_ms"] = h2d["pinned"]
        out["h2d_gbs"] = in_bytes / h2d["pinned"] / 1e6
    out["note"] = ("rank 0, whole shard; h2d_ms / d2h_ms: pinned host memory (ndlqr_HostAlloc), *_pageable: malloc'ed; "
                   "not part of `value`")
    return out, end_to_end, sol


def launch_ranks(gpus):
    """--gpus N > 1 outside a launcher: start N fresh rank processes (one per GPU) with torch.distributed.run and
    exit with their status. Nothing in this process has touched the GPU (no torch import, no HIP call)."""
    sock = socket.socket()
    sock.bind(("127.0.0.1", 0))
    port = sock.getsockname()[1]
    sock.close()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC (RCCL across processes on this pool)
    env.setdefault("OMP_NUM_THREADS", "1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    log("--gpus %d without WORLD_SIZE: launching %d ranks: %s" % (gpus, gpus, " ".join(cmd[1:])))
    sys.exit(subprocess.run(cmd, env=env).returncode)


def roofline_object(prof, steps, schedule, n, m, N, batch, flags, solves_per_s_per_gpu):
    """The `roofline` object of one workload from its per-kernel HIP-event profile: head = dominant kernel kind
    (algorithmic bytes / useful flops of the IMPLEMENTED schedule over the measured launch time), `kernels` = the
    other kinds, `step` = the whole step, `traffic` = PMC-measured HBM bytes of the committed profile of this tree."""
    used = {k: v for k, v in prof.items() if v[1] > 0}
    model = rf.model_for(schedule, n, m, N)
    traffic, traffic_src = committed_traffic(n, m, N, batch, flags)
    issue, issue_src = committed_issue(n, m, N, batch, flags)
    kernels = {}
    for slot, (ms, launches) in used.items():
        avg_ms = ms / launches
        entry = {"avg_launch_ms": avg_ms, "launches_per_step": launches / steps,
                 "ms_per_step": ms / steps}
        if model and slot in model:
            entry.update(rf.kernel_roofline(model[slot], batch, avg_ms, launches / steps))
            assert 0.0 < entry["frac"] <= 1.0, (slot, entry)
        entry["traffic"] = traffic.get(slot)
        if slot in issue:
            # the time the launch's instructions take on the fp64 pipe of the SIMDs (vector and fp64 matrix-core
            # instructions execute one after the other on gfx950): the bound of a kernel that is not waiting for memory
            i = issue[slot]
            entry["fp64_pipe"] = {"vector_per_wavefront": round(i["vector_per_wavefront"], 1),
                                  "matrix_per_wavefront": round(i["matrix_per_wavefront"], 1),
                                  "exec_ms_per_step": i["exec_ms_per_solve"],
                                  "frac": i["exec_ms_per_solve"] / (ms / steps), "source": issue_src}
        kernels[slot] = entry
    dom = max(kernels, key=lambda k: kernels[k]["ms_per_step"])
    roofline = dict(kernels[dom])
    roofline["kernel"] = dom
    roofline["schedule"] = schedule
    if "frac" not in roofline:
        # strict / KEEP schedules stream the whole factor array level by level: SURVEY 8(d) model (B)
        # is what they execute
        b = rf.model_b_bytes(n, m, N) * batch
        gbs = b / (sum(k["ms_per_step"] for k in kernels.values()) * 1e-3) / 1e9
        roofline.update(bound="hbm", achieved=gbs, peak=rf.HBM_PEAK_GBS, unit="GB/s", frac=gbs / rf.HBM_PEAK_GBS,
                        note="whole step against SURVEY.md 8(d) model (B) level-streaming bytes")
    roofline["traffic_source"] = traffic_src
    if model:
        sb = sum(v["bytes"] for v in model.values()) * batch
        sf = sum(v["flops"] for v in model.values()) * batch
        dev_ms = sum(k["ms_per_step"] for k in kernels.values())
        roofline["step"] = {
            "algorithmic_bytes": sb, "useful_flops": sf, "kernel_ms": dev_ms,
            "hbm_frac": sb / (dev_ms * 1e-3) / 1e9 / rf.HBM_PEAK_GBS,
            "fp64_frac": sf / (dev_ms * 1e-3) / 1e12 / rf.FP64_PEAK_TFLOPS,
            # inputs read once + solution written once (SURVEY 8(d) floor (A)): what a perfectly
            # fused solver would move -- the step's distance from the HBM roof of the PROBLEM
            "compulsory_bytes": rf.compulsory_bytes(n, m, N) * batch,
            "compulsory_hbm_frac": rf.compulsory_bytes(n, m, N) * batch / (dev_ms * 1e-3) / 1e9 / rf.HBM_PEAK_GBS,
            "traffic": sum(kernels[k]["traffic"] * kernels[k]["launches_per_step"] for k in kernels)
            if all(kernels[k]["traffic"] for k in kernels) else None,
        }
    roofline["kernels"] = {k: v for k, v in kernels.items() if k != dom}
    # the reference's dense schedule priced at this throughput (> peak: the bytes are not moved)
    roofline["vs_level_streaming_model"] = {
        "model_b_bytes_per_solve": rf.model_b_bytes(n, m, N),
        "equivalent_gbs_per_gpu": rf.model_b_bytes(n, m, N) * solves_per_s_per_gpu / 1e9,
        "model_b_flops_per_solve": rf.model_b_flops(n, m, N)}
    return roofline


def profile_pass(rslqr_amd, bs, flags, steps):
    """Per-kernel durations: `steps` more solves with a HIP-event pair around every launch on the launch stream
    (events force eager launches, so they cannot sit inside the graph-replayed region; the kernels and their
    inputs are identical). Returns ({slot: (ms, launches)}, wall seconds)."""
    bs.set_flags(flags | rslqr_amd.FLAG_PROFILE)
    bs.solve()
    bs.profile_reset()
    t0 = time.perf_counter()
    for _ in range(steps):
        bs.solve_async()
    bs.synchronize()
    wall = time.perf_counter() - t0
    prof = bs.profile()
    bs.set_flags(flags)
    return prof, wall


def config_leg(rslqr_amd, n, m, N, batch, device, steps, seed0=1, json_path=None, cpu_sample=0):
    """One more BASELINE configuration on this GPU, measured like the headline workload on a bounded number of
    steps: spin-up, `steps` graph-replayed solves between synchronisations, the per-kernel profile pass, the
    device-side KKT residual of every problem. json_path: the problem of a reference JSON fixture (batch 1)."""
    t_leg = time.perf_counter()
    bs = rslqr_amd.BatchSolver(n, m, N, batch, device=device)
    out = {"workload": "nx=%d nu=%d N=%d batch=%d, fp64, factor+solve per step" % (n, m, N, batch)}
    try:
        L = rslqr_amd.lib()
        soln = None
        if json_path:
            import ctypes as C
            prob = L.ndlqr_ReadLQRProblemJSONFile(json_path.encode())
            if not prob:
                return {"error": "cannot read " + json_path}
            arr = (C.POINTER(rslqr_amd.LQRProblem) * 1)(prob)
            err = L.ndlqr_InitializeBatch(bs.h, arr, 1)
            L.ndlqr_FreeLQRProblem(prob)
            if err:
                return {"error": "ndlqr_InitializeBatch: %d" % err}
            soln = np.asarray(json.load(open(json_path))["soln"], dtype=np.float64).reshape(-1)
            out["data"] = os.path.basename(json_path)
        else:
            bs.initialize_synthetic(seed0)
            out["data"] = "synthetic, seeds %d..%d" % (seed0, seed0 + batch - 1)
        out["setup_s"] = time.perf_counter() - t_leg
        if bs.solve() != 0:
            return {"error": "solve failed: %s" % L.ndlqr_hip_last_error().decode()}
        t0 = time.perf_counter()  # spin-up: ~50 ms of load, at least two solves per buffer set
        spin = 0
        while spin < 4 or ((time.perf_counter() - t0) < 0.05 and spin < 200):
            bs.solve_async()
            bs.solve_async()
            bs.synchronize()
            spin += 2
        t0 = time.perf_counter()
        for _ in range(steps):
            bs.solve_async()
        bs.synchronize()
        ms = (time.perf_counter() - t0) / steps * 1e3
        schedule = bs.schedule()
        out.update(steps=steps, ms_per_step=ms, solves_per_s=batch / (ms * 1e-3), schedule=schedule,
                   pipeline_depth=bs.pipeline_depth(), cholesky_failures=bs.cholesky_failures())
        lat = []
        for _ in range(min(steps, 20)):  # one solve at a time: launch -> synchronise, host clock and device events
            t0 = time.perf_counter()
            bs.solve()
            lat.append(((time.perf_counter() - t0) * 1e3, bs.solve_ms()))
        out["one_solve_in_flight_ms"] = {"host_median": sorted(x[0] for x in lat)[len(lat) // 2],
                                         "device_median": sorted(x[1] for x in lat)[len(lat) // 2]}
        prof, _ = profile_pass(rslqr_amd, bs, 0, min(steps, 10))
        roof = roofline_object(prof, min(steps, 10), schedule, n, m, N, batch, 0, batch / (ms * 1e-3))
        roof.pop("vs_level_streaming_model", None)
        out["roofline"] = roof
        res, bn = bs.kkt_residuals()
        out["kkt_residual_rel_max"] = float((res / np.maximum(1.0, bn)).max())
        if soln is not None:
            x = bs.solution(0)
            out["err_vs_fixture_soln_l2"] = float(np.linalg.norm(x - soln))  # the reference's bar: < 1e-6
            out["soln_norm"] = float(np.linalg.norm(soln))
        if cpu_sample:
            probs = [rslqr_amd.generate_synthetic(n, m, N, seed0 + p) for p in range(cpu_sample)]
            base, worst = cpu_baseline(n, m, N, probs, [bs.solution(p) for p in range(cpu_sample)])
            out["cpu_baseline"] = base
            out["parity_rel_err_vs_cpu"] = worst
            out["speedup_vs_cpu"] = out["solves_per_s"] / base["value"]
        out["leg_s"] = time.perf_counter() - t_leg
        return out
    finally:
        bs.close()


def dropin_leg(rslqr_amd, json_path, reps=200, with_reference=True):
    """The reference's own call sequence on one of its JSON fixtures, per call, host clock (what a caller of the drop-in
    sees; src/solve.h:20-32, examples/importexample/main.c:5-27): ndlqr_InitializeWithLQRProblem + ndlqr_Solve +
    ndlqr_CopySolution, steady state. `default`: the first solve of the solver is profiled, the timed ones replay one
    captured graph (copies up, launch chain, copy down); `profiling_every_solve`: ndlqr_SetDeviceProfiling(solver, 1),
    eager launches with an event pair per kernel (the reference's always-on profiler; round 3's default);
    `mirroring_the_factorisation`: ndlqr_SetFactorMirroring(solver, 1), every solve keeps the factor array and brings it
    down into solver->fact (what the reference's ndlqr_Solve leaves there)."""
    import ctypes as C
    L = rslqr_amd.lib()
    prob = L.ndlqr_ReadLQRProblemJSONFile(json_path.encode())
    if not prob:
        return {"error": "cannot read " + json_path}
    n = prob.contents.lqrdata[0].contents.nstates
    m = prob.contents.lqrdata[0].contents.ninputs
    N = prob.contents.nhorizon
    soln = np.asarray(json.load(open(json_path))["soln"], dtype=np.float64).reshape(-1)
    out = {"workload": "%s (%d,%d,%d) x 1, ndlqr_InitializeWithLQRProblem + ndlqr_Solve + ndlqr_CopySolution per call"
                       % (os.path.basename(json_path), n, m, N), "reps": reps}
    x = np.zeros(soln.size)
    xp = x.ctypes.data_as(C.POINTER(C.c_double))
    for mode, prof in (("default", None), ("profiling_every_solve", 1), ("mirroring_the_factorisation", None)):
        solver = L.ndlqr_NewNdLqrSolver(n, m, N)
        if prof is not None:
            L.ndlqr_SetDeviceProfiling(solver, prof)
        if mode == "mirroring_the_factorisation":  # solver->fact left behind by every solve, like the reference's
            L.ndlqr_SetFactorMirroring(solver, 1)
        t = np.zeros((reps + 5, 3))
        rc = 0
        for i in range(reps + 5):
            t0 = time.perf_counter()
            rc |= L.ndlqr_InitializeWithLQRProblem(prob, solver)
            t1 = time.perf_counter()
            rc |= L.ndlqr_Solve(solver)
            t2 = time.perf_counter()
            L.ndlqr_CopySolution(solver, xp)
            t[i] = (t1 - t0, t2 - t1, time.perf_counter() - t2)
        t = np.sort(t[5:] * 1e3, axis=0)
        med = t[len(t) // 2]
        out[mode] = {"initialize_ms": float(med[0]), "solve_ms": float(med[1]), "copy_ms": float(med[2]),
                     "call_sequence_ms": float(med.sum()), "solve_ms_min": float(t[0][1]),
                     "device_ms": float(solver.contents.solve_time_ms), "rc": int(rc),
                     "err_vs_fixture_soln_l2": float(np.linalg.norm(x - soln))}
        L.ndlqr_FreeNdLqrSolver(solver)
    L.ndlqr_FreeLQRProblem(prob)
    if with_reference:
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import support
        if support.have_reference():
            ref = support.Reference()
            p, _ = support.load_json_problem(json_path)
            flat = [np.ascontiguousarray(a[None]) for a in p.arrays()]
            args = [a.ctypes.data_as(support.dp) for a in flat]
            cores = host_cores()
            refms = {}
            for thr in sorted({1, min(4, cores), cores}):
                ref.L.ref_bench(p.n, p.m, p.N, 1, 3, *args, thr)
                refms["%d_threads" % thr] = ref.L.ref_bench(p.n, p.m, p.N, 1, 20, *args, thr) / 20
            out["reference_ndlqr_Solve_ms"] = refms  # the reference's ndlqr_Solve alone (oracle/_ref), same host
    return out


def config4_leg(rslqr_amd, sharding, dist, torch, rank, world, device, backend, batch, steps, barrier):
    """BASELINE config 4 in its real form, on EVERY rank of an N > 1 run: (12,4,1024) with `batch` problems per GPU
    (512 x 8 = 4096 at --gpus 8), batch-sharded with global seeds, timed like the headline (barriers, MAX over ranks)
    and once more with every shard's solutions gathered after each step. Returns the dict on rank 0, None elsewhere."""
    n, m, N = 12, 4, 1024
    tdev = "cuda" if backend == "nccl" else "cpu"
    bs = rslqr_amd.BatchSolver(n, m, N, batch, device=device)
    try:
        bs.initialize_synthetic(sharding.shard_seed0(rank, batch))
        for _ in range(3):
            bs.solve_async()
            bs.solve_async()
            bs.synchronize()
        elapsed = sharding.timed_region(bs, steps, 2, barrier)
        fails = bs.cholesky_failures()
        schedule = bs.schedule()
        kres, kbn = bs.kkt_residuals()
        kkt = float((kres / np.maximum(1.0, kbn)).max())
        local = bs.solutions()
        if backend == "nccl":
            send = torch.empty((batch, bs.nvars), dtype=torch.float64, device="cuda")
            recv = torch.empty((world * batch, bs.nvars), dtype=torch.float64, device="cuda")

            def do_gather():
                bs.solutions_to_device(send.data_ptr())
                bs.synchronize()
                dist.all_gather_into_tensor(recv, send)
                torch.cuda.synchronize()
                return recv
        else:
            def do_gather():
                return sharding.gather_solutions(bs.solutions())
        do_gather()
        gsteps = min(steps, 10)
        g_elapsed, last = sharding.timed_region_with_gather(bs, gsteps, barrier, do_gather)
        mine = last[rank * batch:(rank + 1) * batch]
        mine = mine.cpu().numpy() if hasattr(mine, "cpu") else np.asarray(mine)
        intact = bool(np.array_equal(mine, local))
        per_rank = sharding.all_over_ranks(elapsed, device=tdev)
        red = sharding.max_over_ranks([elapsed, float(fails), kkt, g_elapsed, 0.0 if intact else 1.0], device=tdev)
        if rank != 0:
            return None
        total = batch * world * steps
        return {"workload": "nx=12 nu=4 N=1024 batch=%d per GPU x %d GPUs = %d problems, fp64, factor+solve per step"
                            % (batch, world, batch * world),
                "is_baseline_config4": bool(batch * world == 4096 and world == 8),
                "steps": steps, "value": total / red[0], "unit": "solves/s", "ms_per_step": red[0] / steps * 1e3,
                "elapsed_s_per_rank": per_rank, "elapsed_minmax_s": [min(per_rank), max(per_rank)],
                "schedule": schedule, "cholesky_failures": int(red[1]), "kkt_residual_rel_max": red[2],
                "gather": {"steps": gsteps, "value_incl_gather": batch * world * gsteps / red[3],
                           "ms_per_step_incl_gather": red[3] / gsteps * 1e3,
                           "bytes_per_rank_per_step": 8 * batch * bs.nvars,
                           "every_rank_found_its_shard_intact": red[4] == 0.0}}
    finally:
        bs.close()


def multi_rhs_mode(rslqr_amd, n, m, N, device, seed0, nrhs=1024):
    """ndlqr_SolveBatchMultiRhs: ONE problem, `nrhs` right-hand sides against its one kept factorisation (SURVEY 8f-2
    "multiple right-hand sides"; the reference's NdData holds a single one): device time of the solve kernels, and the whole
    call with its (pageable) transfers."""
    os.environ["NDLQR_TREE"] = "0"  # (the level-per-launch schedule keeps the compact records at batch 1 too)
    try:
        bs = rslqr_amd.BatchSolver(n, m, N, 1, device=device, flags=rslqr_amd.FLAG_KEEP_RECORDS)
    finally:
        del os.environ["NDLQR_TREE"]
    try:
        bs.initialize_synthetic(seed0)
        if bs.solve() != 0 or bs.schedule() != "reduced-compact-records":
            return {"error": "no compact records for this shape (%s)" % bs.schedule()}
        rng = np.random.default_rng(5)
        q, d = rng.standard_normal((nrhs, 1, N, n)), 0.1 * rng.standard_normal((nrhs, 1, N, n))
        r, x0 = rng.standard_normal((nrhs, 1, N, m)), rng.standard_normal((nrhs, 1, n))
        out = np.empty((nrhs, 1, bs.nvars))
        bs.solve_multi_rhs(q, r, d, x0, out=out)
        best, wall = 1e9, 1e9
        for _ in range(3):
            t0 = time.perf_counter()
            bs.solve_multi_rhs(q, r, d, x0, out=out)
            wall = min(wall, (time.perf_counter() - t0) * 1e3)
            best = min(best, bs.solve_ms())
        # one of them against a plain solve of the same problem with that right-hand side
        chk = rslqr_amd.BatchSolver(n, m, N, 1, device=device)
        try:
            chk.initialize_synthetic(seed0)
            chk.set_rhs_flat(q[7], r[7], d[7], x0[7])
            chk.solve()
            ref = chk.solutions()[0]
        finally:
            chk.close()
        # u of knot 0 of every right-hand side alone (ndlqr_SolveBatchMultiRhsSlices)
        u0 = np.empty((nrhs, 1, 1, m))
        sel = (0, 1, rslqr_amd.SOLN_INPUT)
        bs.solve_multi_rhs(q, r, d, x0, out=u0, selection=sel)
        best_u0, wall_u0 = 1e9, 1e9
        for _ in range(3):
            t0 = time.perf_counter()
            bs.solve_multi_rhs(q, r, d, x0, out=u0, selection=sel)
            wall_u0 = min(wall_u0, (time.perf_counter() - t0) * 1e3)
            best_u0 = min(best_u0, bs.solve_ms())
        return {"workload": "1 problem x %d right-hand sides" % nrhs, "kernel_ms": best, "solves_per_s": nrhs / best * 1e3,
                "call_ms_incl_pageable_transfers": wall,
                "rel_err_vs_plain_solve": float(np.linalg.norm(out[7, 0] - ref) / np.linalg.norm(ref)),
                "u0_alone": {"kernel_ms": best_u0, "solves_per_s": nrhs / best_u0 * 1e3,
                             "call_ms_incl_pageable_transfers": wall_u0,
                             "equals_whole_vectors": bool(np.array_equal(u0[:, 0, 0, :], out[:, 0, 2 * n:2 * n + m]))}}
    finally:
        bs.close()


def time_mode(rslqr_amd, n, m, N, batch, device, seed0, flags, steps, rhs_only=False):
    """ms per step of one more mode of the same workload (own solver, same synthetic problems)."""
    bs = rslqr_amd.BatchSolver(n, m, N, batch, device=device, flags=flags)
    try:
        bs.initialize_synthetic(seed0)
        if bs.solve() != 0:
            return None
        t0 = time.perf_counter()
        if rhs_only:
            for _ in range(steps):
                if bs.solve_rhs_only() != 0:
                    return None
        else:
            for _ in range(steps):
                bs.solve_async()
            bs.synchronize()
        ms = (time.perf_counter() - t0) / steps * 1e3
        res, bn = bs.kkt_residuals()
        return {"ms_per_step": ms, "solves_per_s": batch / (ms * 1e-3), "schedule": bs.schedule(),
                "kkt_residual_rel_max": float((res / np.maximum(1.0, bn)).max())}
    finally:
        bs.close()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--spin-up-ms", type=float, default=60.0,
                    help="untimed full solves before the warm-up steps until the device has been under load this long (0: none)")
    ap.add_argument("--nx", type=int, default=12)
    ap.add_argument("--nu", type=int, default=4)
    ap.add_argument("--horizon", type=int, default=256)
    ap.add_argument("--batch", type=int, default=1024, help="problems per GPU")
    ap.add_argument("--flags", type=int, default=0, help="NDLQR_FLAG_* bits (1 strict, 2 generic, 8 keep fact)")
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    ap.add_argument("--no-modes", action="store_true", help="skip the strict / KEEP / rhs-only legs")
    ap.add_argument("--no-gather", action="store_true", help="N>1: skip the solution-gather leg")
    ap.add_argument("--cpu-sample", type=int, default=0, help="problems in the CPU sample (0 = auto)")
    ap.add_argument("--no-transfers", action="store_true", help="skip the transfer / end-to-end legs")
    ap.add_argument("--transfer-leg", action="store_true", help=argparse.SUPPRESS)  # internal: child process of the N=1 run
    ap.add_argument("--no-configs", action="store_true", help="N=1: skip the legs of the other BASELINE configurations")
    ap.add_argument("--config4-batch", type=int, default=512,
                    help="N>1: problems per GPU of the (12,4,1024) leg (512 x 8 GPUs = BASELINE config 4; 0: skip)")
    args = ap.parse_args()

    if args.transfer_leg:
        # Child of the N = 1 run (below): the transfer / end-to-end legs in a process of their own that never imports
        # torch. torch 2.10+rocm7.0 brings its own HIP runtime (torch/lib/libamdhip64.so, 7.0), which every library
        # loaded after it then uses; on it copies and kernels of two streams overlap far less than on the system's
        # ROCm 7.2 runtime that a plain C caller of the library gets (x0-only step 1.72 vs 1.19 ms,
        # tools/e2e_probe.py --with-torch). The parent is idle while this runs.
        import rslqr_amd
        n, m, N, batch = args.nx, args.nu, args.horizon, args.batch
        bs = rslqr_amd.BatchSolver(n, m, N, batch, device=0, flags=args.flags)
        bs.initialize_synthetic(1)
        for _ in range(8):
            bs.solve_async()
        bs.synchronize()
        transfers, end_to_end, _ = transfer_legs(rslqr_amd, bs, n, m, N, batch, 1, min(args.steps, 50))
        bs.close()
        print(json.dumps({"transfers": transfers, "end_to_end": end_to_end}), flush=True)
        return
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        launch_ranks(args.gpus)  # does not return
    if os.environ.get("NDLQR_BENCH_LAUNCH_ONLY"):  # test hook (tests/test_sharding_gloo.py): the launch alone, no GPU
        print(json.dumps({"rank": int(os.environ.get("RANK", "0")), "world": int(os.environ.get("WORLD_SIZE", "1")),
                          "gpus": args.gpus}), flush=True)
        sys.exit(0 if int(os.environ.get("WORLD_SIZE", "1")) == args.gpus else 3)

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
    # (NDLQR_BENCH_FORCE_DIST=1: initialise the process group even for one rank -- exercises the RCCL init, reductions
    #  and all_gather of the N > 1 path on a one-GPU box: tests/test_sharding_gloo.py::test_bench_rccl_path_one_rank)
    distributed = world > 1 or bool(os.environ.get("NDLQR_BENCH_FORCE_DIST"))
    if distributed:
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)
    ranks_seen = dist.get_world_size() if distributed else 1
    if ranks_seen != args.gpus:
        log("ERROR: --gpus %d but the process group has %d ranks" % (args.gpus, ranks_seen))
        if distributed:
            dist.destroy_process_group()
        sys.exit(3)

    import rslqr_amd
    n, m, N, batch = args.nx, args.nu, args.horizon, args.batch
    steps = args.steps
    big = rf.model_b_flops(n, m, N) > 2e9
    if big and args.steps == 100:
        steps = 10  # (64,16,512) x 256: ~50 ms per step
    bs = rslqr_amd.BatchSolver(n, m, N, batch, device=local_rank, flags=args.flags)
    seed0 = sharding.shard_seed0(rank, batch)  # global problem g has seed 1 + g

    log("rank %d: generating + uploading %d synthetic problems" % (rank, batch))
    bs.initialize_synthetic(seed0)
    log("rank %d: warm-up" % rank)

    def barrier():
        if distributed:
            dist.barrier()
        torch.cuda.synchronize()

    # Device spin-up, untimed and BEFORE the W warm-up steps: the first solves carry one-time work (capture of the
    # launch sequence for both buffer sets of the pipeline, lazy allocation of the second set), and the GPU needs
    # some tens of milliseconds of load to reach its sustained clocks -- W = 5 steps are 3 ms at this workload, and
    # a 20-step region started right behind them measured 0.745 ms/step where every later one measured 0.65-0.67
    # (tools/overlap_probe.py style loop). Full solves, reported as `spin_up`.
    spin_steps, spin_t0 = 0, time.perf_counter()
    while args.spin_up_ms > 0 and (time.perf_counter() - spin_t0) * 1e3 < args.spin_up_ms and spin_steps < 400:
        for _ in range(4):
            bs.solve_async()
        bs.synchronize()
        spin_steps += 4
    spin_ms = (time.perf_counter() - spin_t0) * 1e3

    # Timed region: the product path as shipped (launch sequence replayed as a hipGraph).
    elapsed = sharding.timed_region(bs, steps, args.warmup, barrier)
    log("rank %d: %d steps in %.3f s" % (rank, steps, elapsed))
    fails = bs.cholesky_failures()
    schedule = bs.schedule()
    # The same K steps strictly stream-ordered (pipeline depth 1: a step starts when the previous one has
    # finished), next to the default two-deep pipeline of `value` (include/ndlqr_hip.h)
    depth = bs.pipeline_depth()
    elapsed_ordered = None
    if depth > 1:
        bs.set_pipeline_depth(1)
        elapsed_ordered = sharding.timed_region(bs, steps, 2, barrier)
        bs.set_pipeline_depth(depth)

    # Per-kernel durations for the roofline object: the SAME K steps once more with a HIP-event
    # pair around every launch on the launch stream. (Events force eager launches, so they cannot
    # sit inside the graph-replayed region above; the kernels and their inputs are identical.)
    prof, elapsed_profiled = profile_pass(rslqr_amd, bs, args.flags, steps)

    # per-solve device times (HIP events around the replayed launch sequence, one solve at a time):
    # median and minimum next to the mean of the timed region
    per_solve = []
    for _ in range(min(steps, 50)):
        bs.solve()
        per_solve.append(bs.solve_ms())
    per_solve.sort()
    dev_median, dev_min = per_solve[len(per_solve) // 2], per_solve[0]

    # every problem of the shard, checked on the device against its raw data (outside the timed
    # region): worst ||K z - b|| / max(1, ||b||) -- SURVEY.md 8(d) "KKT residual of every problem"
    kres, kbn = bs.kkt_residuals()
    kkt_worst = float((kres / np.maximum(1.0, kbn)).max())

    # ---- transfers and the end-to-end MPC step, timed on their own: N = 1 only, in a child process without torch
    #      (see --transfer-leg above); the GPU is otherwise idle meanwhile
    local_sol = bs.solutions()
    transfers, end_to_end = None, None
    if world == 1 and not args.no_transfers:
        log("transfer / end-to-end legs (child process on the system HIP runtime)")
        cmd = [sys.executable, os.path.abspath(__file__), "--transfer-leg", "--steps", str(steps), "--nx", str(n),
               "--nu", str(m), "--horizon", str(N), "--batch", str(batch), "--flags", str(args.flags)]
        try:
            proc = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
            lines = [l for l in proc.stdout.splitlines() if l.startswith("{")]
            if proc.returncode == 0 and lines:
                leg = json.loads(lines[-1])
                transfers, end_to_end = leg["transfers"], leg["end_to_end"]
                transfers["runtime"] = end_to_end["runtime"] = "child process without torch: system ROCm HIP runtime"
            else:
                log("transfer leg failed (rc %d): %s" % (proc.returncode, proc.stderr[-500:]))
        except subprocess.TimeoutExpired:
            log("transfer leg timed out")

    # ---- N>1: the same steps with every shard's solutions gathered after each
    gather = None
    if distributed and not args.no_gather:
        gsteps = min(steps, 20)
        if backend == "nccl":
            send = torch.empty((batch, bs.nvars), dtype=torch.float64, device="cuda")
            recv = torch.empty((world * batch, bs.nvars), dtype=torch.float64, device="cuda")

            def do_gather():
                bs.solutions_to_device(send.data_ptr())  # pack kernel on the solver's stream
                bs.synchronize()
                dist.all_gather_into_tensor(recv, send)   # RCCL over xGMI
                torch.cuda.synchronize()
                return recv
        else:
            def do_gather():
                return sharding.gather_solutions(bs.solutions())
        do_gather()
        g_elapsed, last = sharding.timed_region_with_gather(bs, gsteps, barrier, do_gather)
        mine = last[rank * batch:(rank + 1) * batch]
        mine = mine.cpu().numpy() if hasattr(mine, "cpu") else np.asarray(mine)
        gather_ok = bool(np.array_equal(mine, local_sol))
        gather = {"steps": gsteps, "elapsed_s": g_elapsed, "own_shard_intact": gather_ok,
                  "bytes_per_rank_per_step": 8 * batch * bs.nvars}

    tdev = "cuda" if backend == "nccl" else "cpu"
    elapsed_per_rank = sharding.all_over_ranks(elapsed, device=tdev) if distributed else [elapsed]
    red = [elapsed, float(fails), kkt_worst]
    if elapsed_ordered is not None:
        elapsed_ordered = sharding.max_over_ranks([elapsed_ordered], device="cuda" if backend == "nccl" else "cpu")[0]
    if gather:
        red += [gather["elapsed_s"], 0.0 if gather["own_shard_intact"] else 1.0]
    red = sharding.max_over_ranks(red, device="cuda" if backend == "nccl" else "cpu")
    elapsed_max, fails_max, kkt_max = red[0], int(red[1]), red[2]

    # ---- N>1: BASELINE config 4 on every rank ((12,4,1024) x 512 per GPU; the gather included), headline workload only
    headline = (n, m, N, batch, args.flags) == (12, 4, 256, 1024, 0) or bool(os.environ.get("NDLQR_BENCH_FORCE_CONFIG4"))
    config4 = None
    if distributed and world > 1 and headline and not args.no_configs and args.config4_batch > 0:
        log("rank %d: config 4 leg, (12,4,1024) x %d per GPU" % (rank, args.config4_batch))
        bs.close()
        config4 = config4_leg(rslqr_amd, sharding, dist, torch, rank, world, local_rank, backend, args.config4_batch,
                              min(steps, 20), barrier)

    if rank == 0:
        total_solves = batch * world * steps
        value = total_solves / elapsed_max
        step_ms = elapsed_max / steps * 1e3
        roofline = roofline_object(prof, steps, schedule, n, m, N, batch, args.flags, value / world)
        result = {
            "metric": "LQR solves/sec (nx=%d,nu=%d,N=%d,batch=%d per GPU)" % (n, m, N, batch),
            "value": value, "unit": "solves/s", "n_gpus": world, "steps": steps,
            "warmup": args.warmup, "ms_per_step": step_ms,
            "ms_per_solve": 1e3 / value,
            "device_ms_per_step": {"median": dev_median, "min": dev_min, "reps": len(per_solve),
                                   "note": "rank 0, HIP events per solve, one solve in flight"},
            "elapsed_s_per_rank": elapsed_per_rank, "elapsed_minmax_s": [min(elapsed_per_rank), max(elapsed_per_rank)],
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64",
            "data": "synthetic (seeded splitmix64 family of SURVEY.md 8d, time-varying A,B,Q,R)",
            "config": {"workload": "nx=%d nu=%d N=%d batch=%d per GPU, fp64, factor+solve per step"
                                   % (n, m, N, batch),
                       "parallelism": "batch-sharded x%d, no data-path collective" % world,
                       "ranks_seen": ranks_seen, "backend": backend if distributed else None,
                       "flags": args.flags, "schedule": schedule, "cholesky_failures": fails_max,
                       "kkt_residual_rel_max": kkt_max, "csrc_sha": csrc_sha()},
            "roofline": roofline,
            "kernel_ms_note": "second pass of the same %d steps with per-launch HIP events (eager "
                              "launches): %.3f ms/step vs %.3f ms/step in the timed, graph-replayed region"
                              % (steps, elapsed_profiled / steps * 1e3, step_ms),
            "pipeline": {"depth": depth,
                         "note": "consecutive solves of one solver alternate between two output-buffer sets / "
                                 "streams (each solve is complete; inputs resident and shared); depth 1 = a step "
                                 "starts when the previous one has finished",
                         "value_depth1": (total_solves / elapsed_ordered) if elapsed_ordered else None,
                         "ms_per_step_depth1": (elapsed_ordered / steps * 1e3) if elapsed_ordered else None},
            "spin_up": {"steps": spin_steps, "ms": spin_ms,
                        "note": "untimed full solves before the W warm-up steps: one-time work of the first solves and "
                                "the clock ramp of the device; --spin-up-ms 0 disables"},
            "transfers": transfers,
            "end_to_end": end_to_end,
        }
        if gather:
            g_el, g_bad = red[3], red[4]
            result["gather"] = {
                "steps": gather["steps"], "value_incl_gather": batch * world * gather["steps"] / g_el,
                "ms_per_step_incl_gather": g_el / gather["steps"] * 1e3,
                "collective": "all_gather_into_tensor (RCCL)" if backend == "nccl" else "all_gather (gloo, host)",
                "bytes_per_rank_per_step": gather["bytes_per_rank_per_step"],
                "every_rank_found_its_shard_intact": g_bad == 0.0,
                "note": "solve, pack kernel, host sync, all_gather of [batch, nvars] per step; `value` excludes it"}
        if world == 1 and not args.no_modes and args.flags == 0:
            msteps = min(steps, 20)
            result["modes"] = {}
            if not big:  # (at (64,16,512) x 256 the factor array of these two modes is 87 GB)
                log("secondary modes (strict / KEEP_FACT)")
                result["modes"]["strict_fp (flags=1, bit-identical to the reference)"] = \
                    time_mode(rslqr_amd, n, m, N, batch, local_rank, seed0, 1, msteps)
                result["modes"]["keep_fact (flags=8, complete factor array materialised)"] = \
                    time_mode(rslqr_amd, n, m, N, batch, local_rank, seed0, 8, msteps)
            log("secondary mode (rhs-only re-solve)")
            bs.close()  # the second solver of this leg needs the memory at the large-block sizes
            result["modes"]["rhs_only (flags=16 records kept, new right-hand side per step)"] = \
                time_mode(rslqr_amd, n, m, N, batch, local_rank, seed0, 16, msteps, rhs_only=True)
            if not big:
                log("secondary mode (one problem, 1024 right-hand sides)")
                result["modes"]["multi_rhs (flags=16, ndlqr_SolveBatchMultiRhs)"] = \
                    multi_rhs_mode(rslqr_amd, n, m, N, local_rank, seed0)
        if config4 is not None:
            result.setdefault("configs", {})["config4: (12,4,1024) x %d per GPU x %d GPUs" % (args.config4_batch, world)] = config4
        if not args.no_cpu:  # (rank 0 at any N: the other ranks wait at the closing barrier meanwhile)
            cores = host_cores()
            log("cpu_baseline leg on %d cores" % cores)
            count = args.cpu_sample or (3 if big else max(8, min(batch, 8 * cores)))  # ~10-30 s of CPU work
            count = min(count, batch)
            probs = [rslqr_amd.generate_synthetic(n, m, N, seed0 + p) for p in range(count)]
            gpu_sol = [local_sol[p] for p in range(count)]
            base, worst = cpu_baseline(n, m, N, probs, gpu_sol)
            ric = riccati_column()
            if ric:
                base["riccati"] = ric
            result["cpu_baseline"] = base
            result["parity_rel_err_vs_cpu"] = worst
            result["speedup_vs_cpu"] = value / base["value"]
        headline = (n, m, N, batch, args.flags) == (12, 4, 256, 1024, 0)
        if world == 1 and headline and not args.no_configs:
            # the other single-GPU configurations of BASELINE.json, driver-timed in the same run (bounded legs)
            bs.close()
            fixture = os.path.join(ROOT, "tests", "golden", "lqr_prob_256.json")
            result["configs"] = {}
            log("configs legs: the drop-in call sequence on lqr_prob.json and lqr_prob_256.json")
            result["configs"]["config1: lqr_prob.json (6,3,8) x 1, drop-in ndlqr_Solve"] = \
                {"dropin": dropin_leg(rslqr_amd, os.path.join(ROOT, "tests", "golden", "lqr_prob.json"),
                                      with_reference=not args.no_cpu)}
            dropin2 = dropin_leg(rslqr_amd, fixture, with_reference=not args.no_cpu)
            log("configs leg: lqr_prob_256.json x 1")
            result["configs"]["config2: lqr_prob_256.json (6,3,256) x 1"] = \
                config_leg(rslqr_amd, 6, 3, 256, 1, local_rank, 50, json_path=fixture)
            result["configs"]["config2: lqr_prob_256.json (6,3,256) x 1"]["dropin"] = dropin2
            log("configs leg: (12,4,1024) x 512")
            result["configs"]["config4 shard: (12,4,1024) x 512 (one GPU of the 8 x 512 = 4096 job)"] = \
                config_leg(rslqr_amd, 12, 4, 1024, 512, local_rank, 20)
            log("configs leg: (64,16,512) x 256")
            result["configs"]["config5: (64,16,512) x 256"] = \
                config_leg(rslqr_amd, 64, 16, 512, 256, local_rank, 10, cpu_sample=0 if args.no_cpu else 3)
        print(json.dumps(result), flush=True)

    bs.close()
    if distributed:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

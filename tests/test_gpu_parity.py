"""GPU parity: the HIP path (through the C-ABI of include/ndlqr.h) against the CPU oracle on the
same inputs. Tolerances:
  * NDLQR_FLAG_STRICT_FP: bit-exact (np.array_equal) solution AND factor array vs the oracle,
    which itself is bit-identical to the reference's default build.
  * default (fused multiply-add): relative l2 error <= 1e-9 and KKT residual <= 1e-9*max(1,|b|);
    JSON fixtures additionally the reference's own absolute ||x - soln||_2 < 1e-6
    (test/nested_dissection_test.c:277).
"""
import ctypes as C
import os

import numpy as np
import pytest

from support import GOLDEN, Problem, load_json_problem

pytestmark = pytest.mark.gpu

REL_TOL = 1e-9


def synth(ndlqr, n, m, N, seed):
    g = ndlqr.generate_synthetic(n, m, N, seed)
    return Problem(n, m, N, g["A"], g["B"], g["Q"], g["R"], g["q"], g["r"], g["d"], g["x0"])


def _kkt_ok(oracle, prob, x):
    res, bnorm = oracle.kkt_residual(prob, x)
    return res <= 1e-9 * max(1.0, bnorm), (res, bnorm)


def stack(probs):
    return [np.stack([getattr(p, k) for p in probs]) for k in ("A", "B", "Q", "R", "q", "r", "d", "x0")]


SHAPES = [(6, 3, 8), (6, 3, 16), (6, 3, 64), (12, 4, 8), (12, 4, 16), (12, 4, 64), (5, 2, 32), (16, 8, 8),
          (13, 4, 32), (8, 4, 64), (4, 2, 128), (10, 4, 32), (9, 3, 16), (4, 1, 64), (2, 1, 16), (5, 2, 8),
          (3, 1, 2), (1, 1, 4), (7, 9, 16)]


@pytest.mark.parametrize("n,m,N", SHAPES + [(12, 4, 256), (6, 3, 128)])
@pytest.mark.parametrize("flags", ["generic", "default"])
def test_batch_strict_is_bit_exact(ndlqr, oracle, n, m, N, flags):
    """Strict FP: solution AND complete factor array identical to the oracle, for the runtime-sized
    kernels and for the size-specialised knot-based schedule."""
    batch = 3 if N >= 128 else 5
    probs = [synth(ndlqr, n, m, N, 100 + p) for p in range(batch)]
    fl = ndlqr.FLAG_STRICT_FP | ndlqr.FLAG_KEEP_FACT | (ndlqr.FLAG_GENERIC if flags == "generic" else 0)
    bs = ndlqr.BatchSolver(n, m, N, batch, flags=fl)
    bs.initialize_flat(*stack(probs))
    assert bs.solve() == 0
    sol = bs.solutions()
    for p, prob in enumerate(probs):
        z, fact, _, fails = oracle.solve(prob, 1, want_fact=True)
        assert fails == 0
        assert np.array_equal(sol[p], z[: prob.nvars]), "solution differs from oracle (strict)"
        assert np.array_equal(bs.factors(p), fact), "factor array differs from oracle (strict)"
    bs.close()


@pytest.mark.parametrize("n,m,N", SHAPES + [(12, 4, 256), (6, 3, 512)])
@pytest.mark.parametrize("flags", ["generic", "default"])
def test_batch_fast_within_tolerance(ndlqr, oracle, n, m, N, flags):
    batch = 3
    probs = [synth(ndlqr, n, m, N, 7 + p) for p in range(batch)]
    bs = ndlqr.BatchSolver(n, m, N, batch, flags=ndlqr.FLAG_GENERIC if flags == "generic" else 0)
    bs.initialize_flat(*stack(probs))
    assert bs.solve() == 0
    sol = bs.solutions()
    for p, prob in enumerate(probs):
        z, _, _, _ = oracle.solve(prob, 1)
        ref = z[: prob.nvars]
        rel = np.linalg.norm(sol[p] - ref) / np.linalg.norm(ref)
        assert rel <= REL_TOL, rel
        res, bnorm = oracle.kkt_residual(prob, sol[p])
        assert res <= 1e-9 * max(1.0, bnorm), (res, bnorm)
    # a second solve on the resident inputs gives the same answer (no state leaks between solves)
    assert bs.solve() == 0
    assert np.array_equal(bs.solutions(), sol)
    with pytest.raises(RuntimeError):  # factors are only materialised with NDLQR_FLAG_KEEP_FACT
        bs.factors(0)
    bs.close()


@pytest.mark.parametrize("fname", ["lqr_prob.json", "lqr_prob_256.json"])
def test_json_fixture_through_dropin_api(ndlqr, oracle, fname):
    """The reference's canonical caller (examples/importexample/main.c:5-27)."""
    L = ndlqr.lib()
    path = os.path.join(GOLDEN, fname).encode()
    prob = L.ndlqr_ReadLQRProblemJSONFile(path)
    assert prob
    n = prob.contents.lqrdata[0].contents.nstates
    m = prob.contents.lqrdata[0].contents.ninputs
    N = prob.contents.nhorizon
    solver = L.ndlqr_NewNdLqrSolver(n, m, N)
    assert solver
    assert L.ndlqr_InitializeWithLQRProblem(prob, solver) == 0
    L.ndlqr_FreeLQRProblem(prob)  # Initialize deep-copies (test/test_problem.c:22-25)
    assert L.ndlqr_Solve(solver) == 0
    nvars = L.ndlqr_GetNumVars(solver)
    x = np.zeros(nvars)
    assert L.ndlqr_CopySolution(solver, x.ctypes.data_as(C.POINTER(C.c_double))) == nvars
    pyprob, soln = load_json_problem(os.path.join(GOLDEN, fname))
    assert nvars == soln.size
    assert np.linalg.norm(x - soln) < 1e-6  # the reference's own bar
    z, _, _, _ = oracle.solve(pyprob, 1)
    assert np.linalg.norm(x - z[:nvars]) / np.linalg.norm(z[:nvars]) <= REL_TOL
    prof = L.ndlqr_GetProfile(solver)
    assert prof.t_total_ms > 0 and prof.t_leaves_ms + prof.t_shur_ms > 0  # per-kernel events (default)
    view = L.ndlqr_GetSolution(solver)
    assert view.rows == nvars and view.cols == 1
    assert np.array_equal(view.numpy().ravel(), x)
    assert solver.contents.solve_time_ms > 0
    # SolveTwice (test/nested_dissection_test.c:285-313)
    prob = L.ndlqr_ReadLQRProblemJSONFile(path)
    L.ndlqr_ResetNdData(solver.contents.fact)
    assert L.ndlqr_InitializeWithLQRProblem(prob, solver) == 0
    assert L.ndlqr_Solve(solver) == 0
    L.ndlqr_CopySolution(solver, x.ctypes.data_as(C.POINTER(C.c_double)))
    assert np.linalg.norm(x - soln) < 1e-6
    # hipGraph replay instead of per-kernel events: same answer
    assert L.ndlqr_SetDeviceProfiling(solver, 0) == 0
    prob2 = L.ndlqr_ReadLQRProblemJSONFile(path)
    for _ in range(2):
        assert L.ndlqr_InitializeWithLQRProblem(prob2, solver) == 0
        assert L.ndlqr_Solve(solver) == 0
        y = np.zeros(nvars)
        L.ndlqr_CopySolution(solver, y.ctypes.data_as(C.POINTER(C.c_double)))
        assert np.array_equal(y, x)
    L.ndlqr_FreeLQRProblem(prob2)
    assert solver.contents.solve_time_ms > 0
    # factor mirror on demand
    assert L.ndlqr_SyncFactorsToHost(solver) == 0
    _, fact, _, _ = oracle.solve(pyprob, 1, want_fact=True)
    got = solver.contents.fact.contents.numpy()
    assert np.linalg.norm(got - fact) / np.linalg.norm(fact) <= REL_TOL
    L.ndlqr_FreeLQRProblem(prob)
    L.ndlqr_FreeNdLqrSolver(solver)


@pytest.mark.parametrize("n,m,N,batch", [(64, 16, 32, 2), (32, 8, 64, 2), (20, 20, 16, 3), (32, 16, 64, 2),
                                         (16, 16, 128, 3), (48, 16, 16, 2), (72, 8, 8, 1), (80, 16, 4, 1), (16, 300, 4, 1)])
def test_large_blocks_generic_path(ndlqr, oracle, n, m, N, batch):
    """Shapes without a specialised instance (config 5 family, nx=64 nu=16) run the runtime-sized
    kernels; where the blocks fill 16x16 tiles the fast mode puts the Schur update on
    v_mfma_f64_16x16x4_f64 (kernels_mfma.hpp). Strict mode bit-exact, fast mode within tolerance. (Fast mode
    without KEEP: the separator-only schedule up to 128 states -- beyond what the knot-based kernels reach --, the knot-based
    lean schedule only where the inputs are wider than a workgroup.)"""
    probs = [synth(ndlqr, n, m, N, 900 + p) for p in range(batch)]
    # strict + KEEP, fast + KEEP (full Schur passes), fast without KEEP (boundary knots only +
    # back-substitution over the separator records)
    for strict, keep in ((True, True), (False, True), (False, False)):
        bs = ndlqr.BatchSolver(n, m, N, batch, flags=(ndlqr.FLAG_STRICT_FP if strict else 0) |
                               (ndlqr.FLAG_KEEP_FACT if keep else 0))
        bs.initialize_flat(*stack(probs))
        assert bs.solve() == 0
        if not strict and not keep:
            assert bs.schedule() == ("generic-reduced" if n + m <= 256 else "generic-lean")  # (inputs wider than a workgroup)
        sol = bs.solutions()
        for p, prob in enumerate(probs):
            z, fact, _, fails = oracle.solve(prob, 8, want_fact=True)
            ref = z[: prob.nvars]
            if strict:
                assert np.array_equal(sol[p], ref)
                assert np.array_equal(bs.factors(p), fact)
            else:
                assert np.linalg.norm(sol[p] - ref) / np.linalg.norm(ref) <= REL_TOL
                ok, detail = _kkt_ok(oracle, prob, sol[p])
                assert ok, detail
        bs.close()


@pytest.mark.parametrize("n,m,N,batch", [(16, 4, 2, 2), (16, 4, 4, 3), (32, 8, 8, 2), (48, 12, 64, 2), (64, 16, 128, 3),
                                         (64, 16, 512, 1),
                                         # blocks that do not fill their tiles: zero-padded in LDS (PAD instances)
                                         (20, 20, 16, 3), (7, 9, 16, 2), (5, 3, 32, 2), (1, 1, 4, 2), (17, 3, 16, 2),
                                         (33, 5, 8, 2), (50, 10, 64, 2), (63, 1, 16, 2), (64, 15, 16, 1), (3, 1, 2, 2),
                                         # inputs too wide for one wavefront per tile column: the larger workgroups
                                         (64, 200, 8, 1), (48, 150, 8, 1), (32, 100, 8, 2), (16, 60, 16, 2),
                                         # beyond 64 states: five to eight tile columns, one workgroup per CU
                                         (80, 16, 16, 2), (96, 16, 8, 1), (72, 8, 32, 2), (90, 6, 8, 1), (65, 3, 4, 1),
                                         (96, 32, 4, 1), (112, 16, 8, 1), (128, 16, 8, 2), (120, 10, 8, 1), (100, 4, 4, 1),
                                         (113, 7, 4, 1)])
def test_separator_only_schedule_large_blocks(ndlqr, oracle, n, m, N, batch, monkeypatch):
    """Every block size up to 128 states (whose staged [A | B] fits the LDS) takes the separator-only schedule on the matrix cores on ITS OWN block size
    (kernels_reduced_mfma.hpp: one launch per tree level, no factor array), down to a single separator (N = 2);
    blocks that do not fill 16x16 tiles are zero-padded in LDS. (NDLQR_NO_PAD=1: by default a block size below 16
    states without a size-specialised instance runs padded inside the next instance instead, test_padded_shapes.)
    Consecutive solves alternate between the two buffer sets of the pipeline and replay the captured graph: every
    one of them has to reproduce the oracle. A non-positive weight is reported like on every other path."""
    monkeypatch.setenv("NDLQR_NO_PAD", "1")
    probs = [synth(ndlqr, n, m, N, 1300 + p) for p in range(batch)]
    refs = [oracle.solve(prob, 1)[0][: prob.nvars] for prob in probs]
    bs = ndlqr.BatchSolver(n, m, N, batch)
    bs.initialize_flat(*stack(probs))
    for _ in range(4):
        assert bs.solve() == 0
        assert bs.schedule() == "generic-reduced"
        sol = bs.solutions()
        for p, prob in enumerate(probs):
            assert np.linalg.norm(sol[p] - refs[p]) / np.linalg.norm(refs[p]) <= REL_TOL
    res, bnorm = bs.kkt_residuals()
    assert np.all(res <= 1e-9 * np.maximum(1.0, bnorm))
    # strict mode and KEEP_FACT leave the schedule (they need the factor array); where the knot-based separator kernel
    # cannot stage S-bar and the panel in LDS (beyond 82 states, tile-filling blocks beyond 112) it keeps them in global memory
    bs.set_flags(ndlqr.FLAG_KEEP_FACT)
    assert bs.solve() == 0 and bs.schedule() == "generic-keep"
    assert np.linalg.norm(bs.solution(0) - refs[0]) / np.linalg.norm(refs[0]) <= REL_TOL
    bs.close()
    if N >= 4:
        bad = probs[0]
        R = bad.R.copy()
        R[N // 2, 0] = -1.0
        bs = ndlqr.BatchSolver(n, m, N, 1)
        bs.initialize_flat(*[np.asarray(a)[None] for a in (bad.A, bad.B, bad.Q, R, bad.q, bad.r, bad.d, bad.x0)])
        assert bs.solve() == -3
        assert bs.cholesky_failures() >= 1
        bs.close()


@pytest.mark.parametrize("n,m,N,batch", [(144, 16, 8, 2), (130, 5, 4, 1), (160, 16, 16, 1), (150, 10, 8, 2), (200, 8, 4, 1),
                                         (256, 32, 4, 1), (96, 16, 8, 1), (112, 16, 4, 1), (128, 16, 4, 1), (176, 14, 4, 1),
                                         (240, 16, 8, 1)])
def test_blocks_beyond_the_lds(ndlqr, oracle, n, m, N, batch):
    """Block sizes whose S-bar and right-hand-side panel do not fit the LDS of the knot-based separator kernel -- every
    block beyond 128 states in every mode, strict mode and KEEP_FACT beyond ~82 -- run that kernel with both arrays in
    global memory (separator_generic, `scratch`): strict mode bit-identical to the oracle on the solution AND the whole
    factor array, fast mode within tolerance, KEEP_FACT + rhs-only re-solve (the factor read where it lies beyond ~140
    states), a non-positive weight reported. (Round 3: a solve beyond 128 states returned NDLQR_ERR_INVALID.)"""
    probs = [synth(ndlqr, n, m, N, 4100 + p) for p in range(batch)]
    full = [oracle.solve(prob, 8, want_fact=True) for prob in probs]
    # (NDLQR_FLAG_KEEP_RECORDS beyond 128 states: no schedule keeps records there, the factor array is kept instead)
    for flags in (ndlqr.FLAG_STRICT_FP | ndlqr.FLAG_KEEP_FACT, ndlqr.FLAG_KEEP_FACT, ndlqr.FLAG_KEEP_RECORDS, 0):
        bs = ndlqr.BatchSolver(n, m, N, batch, flags=flags)
        bs.initialize_flat(*stack(probs))
        assert bs.solve() == 0
        if flags == 0:
            assert bs.schedule() == ("generic-lean" if n > 128 else "generic-reduced")
        if flags == ndlqr.FLAG_KEEP_RECORDS:
            assert bs.schedule() == ("generic-keep" if n > 128 else "generic-reduced-records")
        sol = bs.solutions()
        for p, prob in enumerate(probs):
            z, fact = full[p][0], full[p][1]
            ref = z[: prob.nvars]
            if flags & ndlqr.FLAG_STRICT_FP:
                assert np.array_equal(sol[p], ref)
                assert np.array_equal(bs.factors(p), fact)
            else:
                assert np.linalg.norm(sol[p] - ref) / np.linalg.norm(ref) <= REL_TOL
                ok, detail = _kkt_ok(oracle, prob, sol[p])
                assert ok, detail
                # (the matrix-core separator of the tile-filling blocks leaves other leftovers than the reference in the
                #  never-read upper triangles of the Cholesky factors: compared where separator_generic runs)
                # (beyond 128 states every block runs padded to the next tile-filling size: no separator_generic there)
                if (flags & ndlqr.FLAG_KEEP_FACT) and not (n > 128 or (n % 16 == 0 and (n + m) % 4 == 0)):
                    got = bs.factors(p)
                    assert np.linalg.norm(got - fact) / np.linalg.norm(fact) <= REL_TOL
        if flags & (ndlqr.FLAG_KEEP_FACT | ndlqr.FLAG_KEEP_RECORDS):  # new right-hand side against what was kept
            other = [synth(ndlqr, n, m, N, 4200 + p) for p in range(batch)]
            mixed = [Problem(n, m, N, a.A, a.B, a.Q, a.R, o.q, o.r, o.d, o.x0) for a, o in zip(probs, other)]
            bs.set_rhs_flat(*[np.stack([getattr(q, f) for q in mixed]) for f in ("q", "r", "d", "x0")])
            assert bs.solve_rhs_only() == 0
            sol = bs.solutions()
            for p, prob in enumerate(mixed):
                ref = oracle.solve(prob, 4)[0][: prob.nvars]
                if flags & ndlqr.FLAG_STRICT_FP:
                    assert np.array_equal(sol[p], ref)
                else:
                    assert np.linalg.norm(sol[p] - ref) / np.linalg.norm(ref) <= REL_TOL
        bs.close()
    if N >= 4:
        bad = probs[0]
        Q = bad.Q.copy()
        Q[N // 2, 1] = -2.0
        bs = ndlqr.BatchSolver(n, m, N, 1)
        bs.initialize_flat(*[np.asarray(a)[None] for a in (bad.A, bad.B, Q, bad.R, bad.q, bad.r, bad.d, bad.x0)])
        assert bs.solve() == -3
        assert bs.cholesky_failures() >= 1
        bs.close()


@pytest.mark.parametrize("n,m,N", [(130, 5, 8), (144, 16, 4), (40, 7, 16), (7, 9, 16)])
def test_dropin_solve_any_block_size(ndlqr, oracle, n, m, N):
    """The reference's call sequence (ndlqr_NewLQRProblem / InitializeLQRData / InitializeWithLQRProblem / ndlqr_Solve /
    ndlqr_CopySolution, src/solve.h:20-32) at block sizes that run zero-padded on the device (beyond 128 states; a
    small bucket shape) or on the runtime-sized schedule: the staged one-graph solve against the oracle, twice."""
    L = ndlqr.lib()
    g = ndlqr.generate_synthetic(n, m, N, 77)
    pyprob = Problem(n, m, N, g["A"], g["B"], g["Q"], g["R"], g["q"], g["r"], g["d"], g["x0"])
    ref = oracle.solve(pyprob, 4)[0][: pyprob.nvars]
    prob = L.ndlqr_NewLQRProblem(n, m, N)
    assert prob
    dptr = lambda a: np.ascontiguousarray(a, dtype=np.float64).ctypes.data_as(C.POINTER(C.c_double))  # noqa: E731
    keep = []
    for k in range(N):
        arrs = [np.ascontiguousarray(g[f][k], dtype=np.float64) for f in ("Q", "R", "q", "r")]
        A = np.ascontiguousarray(g["A"][k]); B = np.ascontiguousarray(g["B"][k]); d = np.ascontiguousarray(g["d"][k])
        keep += arrs + [A, B, d]
        assert L.ndlqr_InitializeLQRData(prob.contents.lqrdata[k], dptr(arrs[0]), dptr(arrs[1]), dptr(arrs[2]), dptr(arrs[3]),
                                         0.0, dptr(A), dptr(B), dptr(d)) == 0
    x0 = np.ascontiguousarray(g["x0"], dtype=np.float64)
    assert L.ndlqr_InitializeLQRProblem(prob, dptr(x0), prob.contents.lqrdata) == 0
    solver = L.ndlqr_NewNdLqrSolver(n, m, N)
    for rep in range(2):
        assert L.ndlqr_InitializeWithLQRProblem(prob, solver) == 0
        assert L.ndlqr_Solve(solver) == 0
        x = np.zeros(pyprob.nvars)
        assert L.ndlqr_CopySolution(solver, x.ctypes.data_as(C.POINTER(C.c_double))) == pyprob.nvars
        assert np.linalg.norm(x - ref) / np.linalg.norm(ref) <= REL_TOL, rep
    L.ndlqr_FreeNdLqrSolver(solver)
    L.ndlqr_FreeLQRProblem(prob)


def test_fuzz_parity_short():
    """Twenty seconds of tools/fuzz_parity.py (random shapes, batches, flags, schedules and call sequences against the
    oracle; a fresh process: it switches NDLQR_TREE between cases) with a fixed seed: no mismatch, no exception."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "tools", "fuzz_parity.py"), "20", "11"], capture_output=True,
                         text=True, timeout=600)
    tail = out.stdout[-1500:] + out.stderr[-1500:]
    assert out.returncode == 0 and "0 failures" in out.stdout, tail


def test_generic_flag_switch_on_specialised_shape(ndlqr, oracle):
    """A context of a size-specialised shape comes with the (smaller) accumulator array of ITS separator-only
    schedule; NDLQR_FLAG_GENERIC sends it through the runtime-sized schedule, which needs its own, larger one:
    reallocated on the first such solve, for both buffer sets of the pipeline, and the captured launch
    sequences that hold the old address are dropped. Switching back and forth keeps every solve right."""
    n, m, N, batch = 12, 4, 64, 5
    probs = [synth(ndlqr, n, m, N, 1500 + p) for p in range(batch)]
    refs = [oracle.solve(prob, 1)[0][: prob.nvars] for prob in probs]
    bs = ndlqr.BatchSolver(n, m, N, batch)
    bs.initialize_flat(*stack(probs))
    for flags, name in ((0, "reduced-tree"), (0, "reduced-tree"), (ndlqr.FLAG_GENERIC, "generic-reduced"),
                        (0, "reduced-tree"), (ndlqr.FLAG_GENERIC, "generic-reduced"),
                        (ndlqr.FLAG_GENERIC, "generic-reduced"), (0, "reduced-tree"), (0, "reduced-tree"),
                        (0, "reduced-tree")):
        bs.set_flags(flags)
        assert bs.solve() == 0
        assert bs.schedule() == name  # (a small batch: the one-launch tree variant of the specialised schedule)
        sol = bs.solutions()
        for p in range(batch):
            assert np.linalg.norm(sol[p] - refs[p]) / np.linalg.norm(refs[p]) <= REL_TOL
    bs.close()


def test_non_spd_block_is_reported(ndlqr):
    """A non-positive R entry: the reference's Cholesky fails silently (src/linalg.c:80-85,
    src/solve.c:189); here the solve returns NDLQR_ERR_NOT_SPD and counts the failure."""
    n, m, N = 12, 4, 16
    g = ndlqr.generate_synthetic(n, m, N, 5)
    g["R"][3, 1] = -1.0
    for flags in (0, ndlqr.FLAG_GENERIC):
        bs = ndlqr.BatchSolver(n, m, N, 1, flags=flags)
        bs.initialize_flat(*[g[k][None] for k in ("A", "B", "Q", "R", "q", "r", "d", "x0")])
        assert bs.solve() == -3
        assert bs.cholesky_failures() >= 1
        bs.close()


def test_non_spd_step_is_reported_one_step_behind(ndlqr):
    """An MPC loop that only ever calls ndlqr_BatchSynchronizePrevious must learn that a step's solve met a
    non-positive pivot before it consumes that step's `soln`: the call that waits for the step returns
    NDLQR_ERR_NOT_SPD (the reference's ndlqr_Solve never tells, src/solve.c:189); later healthy steps return 0 again."""
    n, m, N, batch = 12, 4, 64, 40
    gens = [ndlqr.generate_synthetic(n, m, N, 300 + p) for p in range(batch)]
    flat = {k: np.stack([g[k] for g in gens]) for k in ("A", "B", "Q", "R", "q", "r", "d", "x0")}
    bad = {k: v.copy() for k, v in flat.items()}
    bad["R"][7, 5, 2] = -0.5
    x0 = ndlqr.pinned_empty(flat["x0"].shape); x0[...] = flat["x0"]
    outs = [ndlqr.pinned_empty((batch, (2 * n + m) * N - m)) for _ in range(2)]
    bs = ndlqr.BatchSolver(n, m, N, batch)
    bs.initialize_flat(*[bad[k] for k in ("A", "B", "Q", "R", "q", "r", "d", "x0")])
    assert bs.step_async(None, None, None, x0, outs[0]) == 0      # step 0: non-SPD
    assert bs.step_async(None, None, None, x0, outs[1]) == 0      # step 1: the same inputs
    assert bs.synchronize_previous() == -3                         # waits for step 0
    assert bs.cholesky_failures() >= 1
    assert bs.synchronize() == 0
    bs.initialize_flat(*[flat[k] for k in ("A", "B", "Q", "R", "q", "r", "d", "x0")])
    for s in range(3):
        assert bs.step_async(None, None, None, x0, outs[s & 1]) == 0
        if s >= 1:
            assert bs.synchronize_previous() == 0
    assert bs.synchronize() == 0 and bs.cholesky_failures() == 0
    bs.close()


@pytest.mark.parametrize("n,m,N", [(12, 4, 64), (12, 4, 256), (6, 3, 32), (13, 4, 16), (4, 1, 8), (10, 4, 128),
                                   # runtime-sized separator-only schedule: the compact records (factors) + slots of every separator
                                   (16, 4, 16), (20, 20, 16), (64, 16, 32), (7, 9, 8), (16, 4, 2), (33, 5, 4), (48, 16, 64),
                                   (80, 16, 8), (90, 5, 4), (128, 8, 4), (100, 4, 4)])
def test_resolve_with_records_only(ndlqr, oracle, n, m, N):
    """NDLQR_FLAG_KEEP_RECORDS: the lean fast-mode solve keeps just the separator records and
    factors; new right-hand sides are then solved without the factor array -- also after the
    matrices were replaced and the (graph-replayed) solve ran again. Size-specialised shapes and, up to 64
    states, every other one (records + accumulator slots + W = L^-1 of every separator)."""
    batch = 3
    first = [synth(ndlqr, n, m, N, 300 + p) for p in range(batch)]
    other = [synth(ndlqr, n, m, N, 700 + p) for p in range(batch)]
    bs = ndlqr.BatchSolver(n, m, N, batch, flags=ndlqr.FLAG_KEEP_RECORDS)
    for mats, rhs in ((first, other), (other, first)):
        mixed = [Problem(n, m, N, f.A, f.B, f.Q, f.R, o.q, o.r, o.d, o.x0) for f, o in zip(mats, rhs)]
        bs.initialize_flat(*stack(mats))
        assert bs.solve() == 0
        for p, prob in enumerate(mats):
            ref = oracle.solve(prob, 1)[0][: prob.nvars]
            assert np.linalg.norm(bs.solution(p) - ref) / np.linalg.norm(ref) <= REL_TOL
        bs.set_rhs_flat(*[np.stack([getattr(p, k) for p in mixed]) for k in ("q", "r", "d", "x0")])
        assert bs.solve_rhs_only() == 0
        sol = bs.solutions()
        for p, prob in enumerate(mixed):
            ref = oracle.solve(prob, 1)[0][: prob.nvars]
            assert np.linalg.norm(sol[p] - ref) / np.linalg.norm(ref) <= REL_TOL
            ok, detail = _kkt_ok(oracle, prob, sol[p])
            assert ok, detail
    with pytest.raises(RuntimeError):  # no factor array in this mode
        bs.factors(0)
    bs.close()
    # on the knot-based runtime-sized kernels (inputs wider than a workgroup, blocks beyond 128 states) no schedule keeps
    # records: the flag keeps the factor array there, and the re-solve is the factor-based sweep
    g = ndlqr.generate_synthetic(16, 300, 4, 5)
    o = ndlqr.generate_synthetic(16, 300, 4, 6)
    bs = ndlqr.BatchSolver(16, 300, 4, 1, flags=ndlqr.FLAG_KEEP_RECORDS)
    bs.initialize_flat(*[g[k][None] for k in ("A", "B", "Q", "R", "q", "r", "d", "x0")])
    assert bs.solve() == 0 and bs.schedule() == "generic-keep"
    bs.set_rhs_flat(*[o[k][None] for k in ("q", "r", "d", "x0")])
    assert bs.solve_rhs_only() == 0
    prob = Problem(16, 300, 4, g["A"], g["B"], g["Q"], g["R"], o["q"], o["r"], o["d"], o["x0"])
    ref = oracle.solve(prob, 1)[0][: prob.nvars]
    assert np.linalg.norm(bs.solution(0) - ref) / np.linalg.norm(ref) <= REL_TOL
    bs.close()


@pytest.mark.parametrize("n,m,N", [(12, 4, 256), (6, 3, 512)])
@pytest.mark.parametrize("a_scale,q_scale", [(1.15, 1.0), (1.0, 1e-4), (1.3, 1e-3)])
def test_harder_problem_families(ndlqr, oracle, n, m, N, a_scale, q_scale):
    """Unstable dynamics (A scaled past the unit circle) and weak state costs: less benign than the
    benchmark family. Strict mode stays bit-identical, fast mode (back-substitution, matrix-core
    products) stays within the stated tolerance and at the oracle's own KKT residual level."""
    g = ndlqr.generate_synthetic(n, m, N, 11)
    g["A"] = g["A"] * a_scale
    g["Q"] = g["Q"] * q_scale
    prob = Problem(n, m, N, g["A"], g["B"], g["Q"], g["R"], g["q"], g["r"], g["d"], g["x0"])
    ref = oracle.solve(prob, 8)[0][: prob.nvars]
    ores, obn = oracle.kkt_residual(prob, ref)
    for strict in (True, False):
        bs = ndlqr.BatchSolver(n, m, N, 1, flags=ndlqr.FLAG_STRICT_FP if strict else 0)
        bs.initialize_flat(*[g[k][None] for k in ("A", "B", "Q", "R", "q", "r", "d", "x0")])
        assert bs.solve() == 0
        sol = bs.solutions()[0]
        if strict:
            assert np.array_equal(sol, ref)
        else:
            assert np.linalg.norm(sol - ref) / np.linalg.norm(ref) <= REL_TOL
            res, bn = oracle.kkt_residual(prob, sol)
            assert res / max(1.0, bn) <= 10.0 * ores / max(1.0, obn) + 1e-12
        bs.close()


def hard_problem(ndlqr, n, m, N, seed, a_scale, q_scale, r_scale):
    g = ndlqr.generate_synthetic(n, m, N, seed)
    g["A"] = g["A"] * a_scale
    g["Q"] = g["Q"] * q_scale
    g["R"] = g["R"] * r_scale
    return g, Problem(n, m, N, g["A"], g["B"], g["Q"], g["R"], g["q"], g["r"], g["d"], g["x0"])


HARD_FAMILIES = [(1.15, 1.0, 1.0), (1.0, 1e-4, 1.0), (1.3, 1e-3, 1.0), (1.0, 1.0, 1e-4)]
# (n, m, N, batch, schedule the fast mode must take): the large-block separator-only schedule at one to eight tile
# columns (explicit inverses of the 16 x 16 diagonal blocks, Gram-form pushes), one PAD instance, the zero-padded
# buckets, and the level-per-launch `reduced` schedule at a batch that selects it
HARD_SHAPES = [(64, 16, 64, 1, "generic-reduced"), (32, 8, 128, 1, "generic-reduced"), (96, 16, 16, 1, "generic-reduced"),
               (128, 16, 8, 1, "generic-reduced"), (50, 10, 64, 1, "generic-reduced"), (7, 9, 64, 1, None),
               (11, 3, 64, 1, None), (12, 4, 256, 40, "reduced-fused2"), (8, 4, 256, 40, "reduced"), (6, 3, 256, 48, "reduced"),
               (13, 4, 128, 80, "reduced"), (12, 4, 256, 1, "reduced-tree"), (6, 3, 64, 2, "reduced-tree"),
               (4, 2, 128, 3, "knot-lean"), (5, 2, 64, 3, "knot-lean"), (2, 1, 64, 3, "knot-lean")]


@pytest.mark.parametrize("n,m,N,batch", [(12, 4, 256, 9), (12, 4, 16, 3), (13, 4, 64, 3), (9, 3, 32, 4), (10, 4, 128, 2),
                                         (6, 3, 64, 5), (8, 4, 1024, 2), (11, 3, 64, 2), (12, 8, 16, 2), (15, 2, 32, 2),
                                         (12, 4, 2048, 1)])
@pytest.mark.parametrize("a_scale,q_scale,r_scale", [(1.0, 1.0, 1.0), (1.0, 1.0, 1e-4), (1.3, 1e-3, 1.0)])
def test_resolve_on_compact_records(ndlqr, oracle, n, m, N, batch, a_scale, q_scale, r_scale, monkeypatch):
    """NDLQR_FLAG_KEEP_RECORDS on the level-per-launch default schedule (round 4): the solve keeps its compact records --
    L of the level-0 separators, f_a | f_bb of the upper ones and, in the slack of the level-0 slots, their factors; no
    factor array -- and ndlqr_SolveBatchRhsOnly runs the right-hand-side column alone (rb_forward, rb_forward_top,
    rb_backsub). New q, r, d AND x0 (the fixed state of knot 0), twice, each against the oracle's full solve."""
    monkeypatch.setenv("NDLQR_TREE", "0")
    gs, probs = zip(*[hard_problem(ndlqr, n, m, N, 600 + p, a_scale, q_scale, r_scale) for p in range(batch)])
    bs = ndlqr.BatchSolver(n, m, N, batch, flags=ndlqr.FLAG_KEEP_RECORDS)
    bs.initialize_flat(*stack(probs))
    assert bs.solve() == 0
    assert bs.schedule() == "reduced-compact-records", bs.schedule()
    sol = bs.solutions()
    for p in sorted(set([0, batch - 1])):
        ref = oracle.solve(probs[p], 8)[0][: probs[p].nvars]
        assert np.linalg.norm(sol[p] - ref) / np.linalg.norm(ref) <= REL_TOL
    for rnd in range(2):
        other = [hard_problem(ndlqr, n, m, N, 700 + 50 * rnd + p, a_scale, q_scale, r_scale)[1] for p in range(batch)]
        mixed = [Problem(n, m, N, a.A, a.B, a.Q, a.R, o.q, o.r, o.d, o.x0) for a, o in zip(probs, other)]
        bs.set_rhs_flat(*[np.stack([getattr(q, f) for q in mixed]) for f in ("q", "r", "d", "x0")])
        for rep in range(2):  # (the re-solve leaves the cached factorisation as it found it)
            assert bs.solve_rhs_only() == 0
            sol = bs.solutions()
            for p in sorted(set([0, batch // 2, batch - 1])):
                ref = oracle.solve(mixed[p], 8)[0][: mixed[p].nvars]
                rel = np.linalg.norm(sol[p] - ref) / np.linalg.norm(ref)
                assert rel <= REL_TOL, (rnd, rep, p, rel)
        res, bn = bs.kkt_residuals()
        ores, obn = oracle.kkt_residual(mixed[0], oracle.solve(mixed[0], 8)[0][: mixed[0].nvars])
        assert float((res / np.maximum(1.0, bn)).max()) <= 10.0 * ores / max(1.0, obn) + 1e-11
    # a full solve afterwards still works (same records, same schedule)
    assert bs.solve() == 0 and bs.schedule() == "reduced-compact-records"
    bs.close()


@pytest.mark.parametrize("n,m,N,batch,nrhs", [(12, 4, 256, 1, 40), (12, 4, 64, 3, 7), (6, 3, 32, 2, 5), (13, 4, 128, 1, 9),
                                              (11, 3, 64, 2, 4), (10, 4, 16, 5, 3), (12, 4, 16, 300, 300)])
def test_multiple_right_hand_sides(ndlqr, oracle, n, m, N, batch, nrhs, monkeypatch):
    """ndlqr_SolveBatchMultiRhs (SURVEY 8f-2 "multiple right-hand sides"): nrhs sets of q, r, d, x0 per problem against the
    ONE factorisation a KEEP_RECORDS solve left, every solution against the oracle's full solve of that problem with that
    right-hand side; the plain re-solve and a full solve afterwards still work (the z_sep of the many right-hand sides
    never touch the records). More sets than fit one launch (65 535 right-hand sides) go in chunks."""
    monkeypatch.setenv("NDLQR_TREE", "0")
    probs = [synth(ndlqr, n, m, N, 5100 + p) for p in range(batch)]
    bs = ndlqr.BatchSolver(n, m, N, batch, flags=ndlqr.FLAG_KEEP_RECORDS)
    bs.initialize_flat(*stack(probs))
    assert bs.solve() == 0 and bs.schedule() == "reduced-compact-records"
    first = bs.solutions().copy()
    rng = np.random.default_rng(3)
    q = rng.standard_normal((nrhs, batch, N, n))
    r = rng.standard_normal((nrhs, batch, N, m))
    d = 0.1 * rng.standard_normal((nrhs, batch, N, n))
    x0 = rng.standard_normal((nrhs, batch, n))
    sol = bs.solve_multi_rhs(q, r, d, x0)
    assert sol.shape == (nrhs, batch, bs.nvars)
    checks = [(0, 0), (nrhs - 1, batch - 1), (nrhs // 2, batch // 2)] if nrhs * batch > 60 else \
        [(j, p) for j in range(nrhs) for p in range(batch)]
    for j, p in checks:
        a = probs[p]
        prob = Problem(n, m, N, a.A, a.B, a.Q, a.R, q[j, p], r[j, p], d[j, p], x0[j, p])
        ref = oracle.solve(prob, 4)[0][: prob.nvars]
        rel = np.linalg.norm(sol[j, p] - ref) / np.linalg.norm(ref)
        assert rel <= REL_TOL, (j, p, rel)
    # slices of every solution, computed alone (ndlqr_SolveBatchMultiRhsSlices): bit for bit those of the whole vectors
    zb = 2 * n + m
    padded = np.zeros((nrhs, batch, N * zb)); padded[:, :, : bs.nvars] = sol
    Z = padded.reshape(nrhs, batch, N, zb)
    for k0, nk, blocks in ((0, 1, ndlqr.SOLN_INPUT), (min(6, N - 4), 4, 7), (N - 1, 1, ndlqr.SOLN_LAMBDA | ndlqr.SOLN_STATE)):
        cols = ([*range(0, n)] if blocks & 1 else []) + ([*range(n, 2 * n)] if blocks & 2 else []) + ([*range(2 * n, zb)] if blocks & 4 else [])
        got = bs.solve_multi_rhs(q, r, d, x0, selection=(k0, nk, blocks))
        assert got.shape == (nrhs, batch, nk, len(cols))
        assert np.array_equal(got, Z[:, :, k0:k0 + nk, :][:, :, :, cols]), (k0, nk, blocks)
    with pytest.raises(RuntimeError):
        bs.solve_multi_rhs(q, r, d, x0, selection=(N - 1, 2, 7))
    # the resident problem is untouched: its own re-solve and a full solve reproduce the first solution
    assert bs.solve_rhs_only() == 0
    again = bs.solutions()
    assert np.linalg.norm(again - first) / np.linalg.norm(first) <= 1e-12
    assert bs.solve() == 0
    assert np.array_equal(bs.solutions(), first)
    bs.close()
    # without the kept records the call is refused
    bs = ndlqr.BatchSolver(n, m, N, batch)
    bs.initialize_flat(*stack(probs))
    assert bs.solve() == 0
    with pytest.raises(RuntimeError):
        bs.solve_multi_rhs(q[:1], r[:1], d[:1], x0[:1])
    bs.close()


@pytest.mark.parametrize("n,m,N,batch,want", HARD_SHAPES)
@pytest.mark.parametrize("a_scale,q_scale,r_scale", HARD_FAMILIES)
def test_harder_families_large_and_padded_paths(ndlqr, oracle, n, m, N, batch, want, a_scale, q_scale, r_scale):
    """The ill-conditioned families of test_harder_problem_families (plus weak input costs) on the paths that only
    ever saw the benign benchmark family: fast mode <= 1e-9 relative against the oracle (the reference's algorithm in
    the reference's operation order) and a KKT residual within 10x the oracle's own; strict mode, where the shape has
    it at this size, bit-exact. Bar of the reference's own test: test/nested_dissection_test.c:277."""
    gs, probs = zip(*[hard_problem(ndlqr, n, m, N, 11 + p, a_scale, q_scale, r_scale) for p in range(batch)])
    sample = sorted(set([0, batch // 2, batch - 1]))
    refs = {p: oracle.solve(probs[p], 8)[0][: probs[p].nvars] for p in sample}
    bs = ndlqr.BatchSolver(n, m, N, batch)
    bs.initialize_flat(*stack(probs))
    assert bs.solve() == 0
    if want:
        assert bs.schedule() == want, bs.schedule()
    sol = bs.solutions()
    kres, kbn = bs.kkt_residuals()
    worst_o = 0.0
    for p in sample:
        ores, obn = oracle.kkt_residual(probs[p], refs[p])
        worst_o = max(worst_o, ores / max(1.0, obn))
        rel = np.linalg.norm(sol[p] - refs[p]) / np.linalg.norm(refs[p])
        assert rel <= REL_TOL, (p, rel)
        res, bn = oracle.kkt_residual(probs[p], sol[p])
        assert res / max(1.0, bn) <= 10.0 * ores / max(1.0, obn) + 1e-12, (p, res, bn, ores, obn)
    # every member, on the device, against its raw data
    assert float((kres / np.maximum(1.0, kbn)).max()) <= 10.0 * worst_o + 1e-11
    bs.close()
    if n <= 64:  # strict mode: where the knot-based kernels reach (include/ndlqr.h, NDLQR_FLAG_STRICT_FP)
        bs = ndlqr.BatchSolver(n, m, N, 1, flags=ndlqr.FLAG_STRICT_FP)
        bs.initialize_flat(*[gs[0][k][None] for k in ("A", "B", "Q", "R", "q", "r", "d", "x0")])
        assert bs.solve() == 0
        assert np.array_equal(bs.solutions()[0], refs[0])
        bs.close()


@pytest.mark.parametrize("n,m,N,flags", [(12, 4, 256, "records"), (6, 3, 128, "records"), (64, 16, 64, "records"),
                                         (12, 4, 64, "fact"), (20, 6, 32, "fact")])
@pytest.mark.parametrize("a_scale,q_scale,r_scale", HARD_FAMILIES)
def test_harder_families_rhs_only_resolve(ndlqr, oracle, n, m, N, flags, a_scale, q_scale, r_scale):
    """The factor / solve split on the ill-conditioned families: a new right-hand side against the cached separator
    records (fast mode) or the cached factor array, within 1e-9 of the oracle's full solve of that problem."""
    g, prob = hard_problem(ndlqr, n, m, N, 21, a_scale, q_scale, r_scale)
    fl = ndlqr.FLAG_KEEP_RECORDS if flags == "records" else ndlqr.FLAG_KEEP_FACT
    bs = ndlqr.BatchSolver(n, m, N, 2, flags=fl)
    bs.initialize_flat(*[np.stack([g[k]] * 2) for k in ("A", "B", "Q", "R", "q", "r", "d", "x0")])
    assert bs.solve() == 0
    rng = np.random.default_rng(5)
    new = {k: g[k] + rng.standard_normal(g[k].shape) for k in ("q", "r", "d", "x0")}
    bs.set_rhs_flat(*[np.stack([new[k]] * 2) for k in ("q", "r", "d", "x0")])
    assert bs.solve_rhs_only() == 0
    p2 = Problem(n, m, N, g["A"], g["B"], g["Q"], g["R"], new["q"], new["r"], new["d"], new["x0"])
    ref = oracle.solve(p2, 8)[0][: p2.nvars]
    sol = bs.solutions()
    for p in range(2):
        assert np.linalg.norm(sol[p] - ref) / np.linalg.norm(ref) <= REL_TOL
    ores, obn = oracle.kkt_residual(p2, ref)
    res, bn = oracle.kkt_residual(p2, sol[0])
    assert res / max(1.0, bn) <= 10.0 * ores / max(1.0, obn) + 1e-12
    bs.close()


def _solve_in_subprocess(n, m, N, batch, seed, env):
    """Solutions of a fresh process with `env` added to the environment (the tuning variables are
    read once, at context creation)."""
    import subprocess, sys, json
    code = (
        "import sys, json, numpy as np; sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
        "import rslqr_amd as R\n"
        "bs = R.BatchSolver(%d, %d, %d, %d); bs.initialize_synthetic(%d)\n"
        "assert bs.solve() == 0; assert bs.solve() == 0; print(json.dumps(bs.solutions().tolist()))\n"
        % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))), os.path.dirname(os.path.abspath(__file__)),
           n, m, N, batch, seed))
    e = dict(os.environ); e.update(env)
    r = subprocess.run([sys.executable, "-c", code], env=e, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, (env, r.stderr[-2000:])
    return np.array(json.loads(r.stdout.strip().splitlines()[-1]))


def test_env_variants_fast_mode(ndlqr, oracle):
    """The separator-only schedule has two bottom kernels (row-broadcast core, matrix-core core) and
    the one-launch tree form for small batches: each stays within the fast-mode tolerance of the
    oracle (relative l2 <= 1e-9, here ~1e-15). Two solves per process: the second one runs on what
    the first left in the accumulators."""
    n, m, N, batch, seed = 12, 4, 128, 3, 91
    probs = [synth(ndlqr, n, m, N, seed + b) for b in range(batch)]
    ref = np.stack([oracle.solve(p, 1)[0][: p.nvars] for p in probs])
    for env in ({}, {"NDLQR_TREE": "1"}, {"NDLQR_TREE": "0"}, {"NDLQR_ROWBCAST": "0", "NDLQR_TREE": "0"}):
        got = _solve_in_subprocess(n, m, N, batch, seed, env)
        err = np.linalg.norm(got - ref) / np.linalg.norm(ref)
        assert err <= REL_TOL, (env, err)


@pytest.mark.parametrize("n,m,N,batch", [(12, 4, 256, 40), (12, 4, 16, 200), (12, 4, 32, 100), (13, 4, 64, 60), (9, 3, 128, 40),
                                         (11, 3, 64, 48)])
def test_fused_bottom_launch(ndlqr, oracle, n, m, N, batch):
    """NDLQR_FUSE2=1: tree level 2 inside the bottom launch (bottom8_reduced_mc: two wavefronts per eight knots, the
    level-2 separator's slot in LDS, its tail shared by the two wavefronts) -- the default at the (12,4) instance, forced
    here at the others (12 % less traffic, about the same kernel time). Same tolerance as the default; the weak-input-cost family too;
    horizons of 16 and 32 knots exercise the top launch starting at level 3 / the root as a level launch."""
    import subprocess, sys, json
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = (
        "import sys, json, numpy as np; sys.path.insert(0, %r)\n"
        "import rslqr_amd as R\n"
        "n, m, N, batch = %d, %d, %d, %d\n"
        "out = {}\n"
        "for fam, rs in (('benign', 1.0), ('weakR', 1e-4)):\n"
        "    gs = [R.generate_synthetic(n, m, N, 60 + p) for p in range(batch)]\n"
        "    for g in gs: g['R'] *= rs\n"
        "    bs = R.BatchSolver(n, m, N, batch)\n"
        "    bs.initialize_flat(*[np.stack([g[k] for g in gs]) for k in ('A','B','Q','R','q','r','d','x0')])\n"
        "    assert bs.solve() == 0 and bs.solve() == 0\n"
        "    res, bn = bs.kkt_residuals()\n"
        "    out[fam] = dict(schedule=bs.schedule(), sol=bs.solutions()[[0, batch - 1]].tolist(), kkt=float((res / np.maximum(1.0, bn)).max()))\n"
        "    bs.close()\n"
        "print(json.dumps(out))\n" % (root, n, m, N, batch))
    r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, NDLQR_FUSE2="1", NDLQR_TREE="0"),
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    out = json.loads(r.stdout.strip().splitlines()[-1])
    for fam, rs in (("benign", 1.0), ("weakR", 1e-4)):
        assert out[fam]["schedule"] == ("reduced-fused2" if n >= 9 else "reduced"), out[fam]["schedule"]
        assert out[fam]["kkt"] <= 1e-9
        for row, p in enumerate((0, batch - 1)):
            g = ndlqr.generate_synthetic(n, m, N, 60 + p)
            g["R"] = g["R"] * rs
            prob = Problem(n, m, N, g["A"], g["B"], g["Q"], g["R"], g["q"], g["r"], g["d"], g["x0"])
            ref = oracle.solve(prob, 4)[0][: prob.nvars]
            got = np.array(out[fam]["sol"][row])
            assert np.linalg.norm(got - ref) / np.linalg.norm(ref) <= REL_TOL, (fam, p)


@pytest.mark.parametrize("n,m,N,batch", [(12, 4, 256, 40), (12, 4, 1024, 3), (6, 3, 128, 64), (13, 4, 64, 60), (12, 4, 32, 100),
                                         (9, 3, 512, 8)])
def test_more_levels_inside_the_top_launch(ndlqr, oracle, n, m, N, batch):
    """NDLQR_TOP_LEVELS=4 / 5: reduced_top_mc starts one or two levels lower, a wavefront taking the eight or sixteen
    separators of the launch's first levels in turn (a knob kept from the measurement of profiles/r04_top_levels_ab.txt).
    Against the default (three levels) and the oracle; horizons where the request exceeds what the tree has fall back to
    three."""
    seed = 2600
    base = _solve_in_subprocess(n, m, N, batch, seed, {"NDLQR_TREE": "0"})
    for levels in ("4", "5"):
        got = _solve_in_subprocess(n, m, N, batch, seed, {"NDLQR_TREE": "0", "NDLQR_TOP_LEVELS": levels})
        err = np.linalg.norm(got - base, axis=1) / np.linalg.norm(base, axis=1)
        assert err.max() <= REL_TOL, (levels, err.max())
        for b in (0, batch - 1):
            prob = synth(ndlqr, n, m, N, seed + b)
            ref = oracle.solve(prob, 8)[0][: prob.nvars]
            assert np.linalg.norm(got[b] - ref) / np.linalg.norm(ref) <= REL_TOL, (levels, b)


@pytest.mark.parametrize("n,m,N,batch", [(12, 4, 256, 32), (6, 3, 256, 48), (13, 4, 512, 24), (7, 9, 256, 40)])
def test_tree_schedule_fills_the_chip(ndlqr, oracle, n, m, N, batch):
    """The one-launch tree schedule at sizes where its bottom wavefronts (e.g. 32 x 256 / 4 = 2048) spread
    over all eight XCDs with several wavefronts per SIMD -- the regime its hand-off protocol
    (write-through pushes, s_waitcnt vmcnt(0), relaxed agent-scope arrival counter, L1-bypassing slot
    loads) has to be right in. Two solves per process (the second starts from the counters the first
    left and reset), every member against the level-per-launch schedule (whose last three levels, reduced_top_mc,
    use the same hand-off inside a workgroup), some against the oracle. Several block sizes, one of them padded."""
    seed = 2100
    tree = _solve_in_subprocess(n, m, N, batch, seed, {"NDLQR_TREE": "1"})
    flat = _solve_in_subprocess(n, m, N, batch, seed, {"NDLQR_TREE": "0", "NDLQR_ROWBCAST": "0"})
    rowb = _solve_in_subprocess(n, m, N, batch, seed, {"NDLQR_TREE": "0", "NDLQR_ROWBCAST": "1"})  # row-broadcast bottom kernel
    err = np.linalg.norm(rowb - flat, axis=1) / np.linalg.norm(flat, axis=1)
    assert err.max() <= REL_TOL, err.max()
    err = np.linalg.norm(tree - flat, axis=1) / np.linalg.norm(flat, axis=1)
    assert err.max() <= REL_TOL, err.max()
    for b in (0, 13, batch - 1):
        prob = synth(ndlqr, n, m, N, seed + b)
        ref = oracle.solve(prob, 8)[0][: prob.nvars]
        assert np.linalg.norm(tree[b] - ref) / np.linalg.norm(ref) <= REL_TOL


@pytest.mark.parametrize("n,m,N,batch,want", [(12, 4, 256, 8, "reduced-tree"), (6, 3, 256, 30, "reduced-tree"),
                                              (12, 4, 128, 96, "reduced-fused2"), (10, 4, 128, 96, "reduced"),
                                              (64, 16, 32, 6, "generic-reduced")])
def test_repeated_solves_are_bitwise_identical(ndlqr, n, m, N, batch, want):
    """The hand-offs between wavefronts (arrival counters and write-through pushes of the tree schedule, atomic adds of
    the level launches, the two buffer sets of the pipeline) leave no room for run-to-run differences: every accumulator
    element receives its additions in a fixed order, so 150 solves of the same inputs are bit for bit the first one.
    A missing ordering edge would show up here as an occasional stale operand."""
    bs = ndlqr.BatchSolver(n, m, N, batch)
    bs.initialize_synthetic(123)
    assert bs.solve() == 0
    assert bs.schedule() == want
    first = bs.solutions()
    for rep in range(50):
        assert bs.solve_async() == 0 and bs.solve_async() == 0 and bs.solve_async() == 0
        assert bs.synchronize() == 0
        assert np.array_equal(bs.solutions(), first), rep
    res, bn = bs.kkt_residuals()
    assert (res <= 1e-9 * np.maximum(1.0, bn)).all()
    bs.close()


def test_solve_pipeline(ndlqr, oracle):
    """Two-deep solve pipeline (include/ndlqr_hip.h): consecutive asynchronous solves alternate between two
    output-buffer sets / streams. Every solve is complete and identical to a stream-ordered one; replacing
    the inputs waits for the solves in flight; the downloads return the most recent solve."""
    n, m, N, batch = 12, 4, 64, 700  # 700 x 16 bottom wavefronts: the level-per-launch schedule
    first = [ndlqr.generate_synthetic(n, m, N, 3000 + p) for p in range(batch)]
    other = [ndlqr.generate_synthetic(n, m, N, 9000 + p) for p in range(batch)]
    flat = lambda gen: [np.stack([g[k] for g in gen]) for k in ("A", "B", "Q", "R", "q", "r", "d", "x0")]
    bs = ndlqr.BatchSolver(n, m, N, batch)
    assert bs.pipeline_depth() == 2
    bs.initialize_flat(*flat(first))
    for _ in range(5):  # odd number: the last solve ran on the primary set, the one before on the alternate
        assert bs.solve_async() == 0
    assert bs.synchronize() == 0
    sol_a = bs.solutions()
    assert bs.solve_async() == 0  # sixth solve: alternate set
    assert bs.synchronize() == 0
    assert np.array_equal(bs.solutions(), sol_a)
    res, bn = bs.kkt_residuals()
    assert (res <= 1e-9 * np.maximum(1.0, bn)).all()
    # new inputs while two solves are in flight: the upload waits for them, the next solves see the new data
    assert bs.solve_async() == 0 and bs.solve_async() == 0
    bs.initialize_flat(*flat(other))
    assert bs.solve_async() == 0 and bs.solve_async() == 0 and bs.solve_async() == 0
    assert bs.synchronize() == 0
    sol_b = bs.solutions()
    assert bs.set_pipeline_depth(1) == 0  # stream-ordered from here on: same bits
    assert bs.solve() == 0
    assert np.array_equal(bs.solutions(), sol_b)
    bs.initialize_flat(*flat(first))
    assert bs.solve() == 0
    assert np.array_equal(bs.solutions(), sol_a)
    for p in (0, batch // 2, batch - 1):
        g = other[p]
        prob = Problem(n, m, N, g["A"], g["B"], g["Q"], g["R"], g["q"], g["r"], g["d"], g["x0"])
        ref = oracle.solve(prob, 1)[0][: prob.nvars]
        assert np.linalg.norm(sol_b[p] - ref) / np.linalg.norm(ref) <= REL_TOL
    bs.close()


def test_dropin_solve_flags(ndlqr, oracle):
    """ndlqr_Solve runs the batch API's default fast path; ndlqr_SetDeviceFlags selects strict mode
    (bit-identical to the reference's default build) or KEEP_FACT up front; ndlqr_SyncFactorsToHost
    works either way (it re-factors the resident problem when the factor array was not kept)."""
    L = ndlqr.lib()
    pyprob, soln = load_json_problem(os.path.join(GOLDEN, "lqr_prob_256.json"))
    z, fact, _, _ = oracle.solve(pyprob, 1, want_fact=True)
    nvars = soln.size
    for flags in (0, ndlqr.FLAG_STRICT_FP, ndlqr.FLAG_KEEP_FACT, ndlqr.FLAG_STRICT_FP | ndlqr.FLAG_KEEP_FACT):
        prob = L.ndlqr_ReadLQRProblemJSONFile(os.path.join(GOLDEN, "lqr_prob_256.json").encode())
        solver = L.ndlqr_NewNdLqrSolver(pyprob.n, pyprob.m, pyprob.N)
        assert L.ndlqr_SetDeviceFlags(solver, flags) == 0
        assert L.ndlqr_InitializeWithLQRProblem(prob, solver) == 0
        assert L.ndlqr_Solve(solver) == 0
        x = np.zeros(nvars)
        L.ndlqr_CopySolution(solver, x.ctypes.data_as(C.POINTER(C.c_double)))
        if flags & ndlqr.FLAG_STRICT_FP:
            assert np.array_equal(x, z[:nvars])
        else:
            assert np.linalg.norm(x - z[:nvars]) / np.linalg.norm(z[:nvars]) <= REL_TOL
        assert L.ndlqr_SyncFactorsToHost(solver) == 0
        got = solver.contents.fact.contents.numpy()
        if flags & ndlqr.FLAG_STRICT_FP:
            assert np.array_equal(got, fact)
        else:
            assert np.linalg.norm(got - fact) / np.linalg.norm(fact) <= REL_TOL
        L.ndlqr_FreeLQRProblem(prob)
        L.ndlqr_FreeNdLqrSolver(solver)


def test_dropin_solve_leaves_the_factorisation(ndlqr, oracle, monkeypatch):
    """ndlqr_SetFactorMirroring / NDLQR_SOLVE_MIRRORS_FACT=1: ndlqr_Solve ends like the reference's (src/solve.c:120-131),
    with the complete factorisation in solver->fact -- no ndlqr_SyncFactorsToHost; bit-identical in strict mode."""
    L = ndlqr.lib()
    for fname, by_env, flags in (("lqr_prob.json", False, 0), ("lqr_prob_256.json", False, ndlqr.FLAG_STRICT_FP),
                                 ("lqr_prob_256.json", True, 0)):
        pyprob, soln = load_json_problem(os.path.join(GOLDEN, fname))
        z, fact, _, _ = oracle.solve(pyprob, 1, want_fact=True)
        if by_env:
            monkeypatch.setenv("NDLQR_SOLVE_MIRRORS_FACT", "1")
        prob = L.ndlqr_ReadLQRProblemJSONFile(os.path.join(GOLDEN, fname).encode())
        solver = L.ndlqr_NewNdLqrSolver(pyprob.n, pyprob.m, pyprob.N)
        monkeypatch.delenv("NDLQR_SOLVE_MIRRORS_FACT", raising=False)
        if not by_env:
            assert solver.contents.mirror_fact == 0
            assert L.ndlqr_SetFactorMirroring(solver, 1) == 0
        assert solver.contents.mirror_fact == 1
        assert L.ndlqr_SetDeviceFlags(solver, flags) == 0
        for rep in range(2):  # (the second call replays the captured sequence)
            assert L.ndlqr_InitializeWithLQRProblem(prob, solver) == 0
            solver.contents.fact.contents.numpy()[:] = -7.0
            assert L.ndlqr_Solve(solver) == 0
            got = solver.contents.fact.contents.numpy()
            x = np.zeros(soln.size)
            L.ndlqr_CopySolution(solver, x.ctypes.data_as(C.POINTER(C.c_double)))
            if flags & ndlqr.FLAG_STRICT_FP:
                assert np.array_equal(got, fact) and np.array_equal(x, z[: soln.size])
            else:
                assert np.linalg.norm(got - fact) / np.linalg.norm(fact) <= REL_TOL
                assert np.linalg.norm(x - z[: soln.size]) / np.linalg.norm(z[: soln.size]) <= REL_TOL
        L.ndlqr_FreeLQRProblem(prob)
        L.ndlqr_FreeNdLqrSolver(solver)


@pytest.mark.parametrize("n,m,N,batch", [(6, 3, 64, 5), (13, 4, 32, 3), (9, 3, 8, 4), (8, 4, 256, 2), (10, 4, 16, 1),
                                         (12, 4, 16, 7), (6, 3, 1024, 2), (13, 4, 512, 2)])
def test_separator_only_schedules_other_shapes(ndlqr, oracle, n, m, N, batch):
    """The separator-only schedules on the other instances (odd row lengths, k-steps with padding,
    shortest horizon with an upper level): default (small batches pick the tree schedule by
    themselves), one launch per level with either bottom kernel, and the tree schedule, against the
    oracle."""
    seed = 500 + n
    probs = [synth(ndlqr, n, m, N, seed + b) for b in range(batch)]
    ref = np.stack([oracle.solve(p, 1)[0][: p.nvars] for p in probs])
    for env in ({}, {"NDLQR_TREE": "0"}, {"NDLQR_TREE": "1"}, {"NDLQR_ROWBCAST": "0", "NDLQR_TREE": "0"},
                {"NDLQR_ROWBCAST": "1", "NDLQR_TREE": "0"}):
        got = _solve_in_subprocess(n, m, N, batch, seed, env)
        err = np.linalg.norm(got - ref) / np.linalg.norm(ref)
        assert err <= REL_TOL, (env, err)


@pytest.mark.parametrize("n,m,N", [(12, 4, 64), (6, 3, 32), (5, 2, 16), (32, 16, 16), (12, 4, 8), (12, 4, 16),
                                   (12, 4, 256), (12, 4, 1024), (13, 4, 64), (4, 2, 128), (8, 4, 32), (6, 3, 4),
                                   (10, 4, 32), (9, 3, 64), (4, 1, 16), (2, 1, 128)])
@pytest.mark.parametrize("strict", [True, False])
def test_factor_solve_split(ndlqr, oracle, n, m, N, strict):
    """Factor once (KEEP_FACT), then new q, r, d, x0 through the rhs-only sweep: must equal a full
    solve of the problem with the new right-hand side (bit-exact in strict mode)."""
    batch = 3
    first = [synth(ndlqr, n, m, N, 300 + p) for p in range(batch)]
    other = [synth(ndlqr, n, m, N, 700 + p) for p in range(batch)]
    mixed = [Problem(n, m, N, f.A, f.B, f.Q, f.R, o.q, o.r, o.d, o.x0) for f, o in zip(first, other)]
    fl = (ndlqr.FLAG_STRICT_FP if strict else 0) | ndlqr.FLAG_KEEP_FACT
    bs = ndlqr.BatchSolver(n, m, N, batch, flags=fl)
    bs.initialize_flat(*stack(first))
    assert bs.solve() == 0
    bs.set_rhs_flat(*[np.stack([getattr(p, k) for p in mixed]) for k in ("q", "r", "d", "x0")])
    assert bs.solve_rhs_only() == 0
    sol = bs.solutions()
    for p, prob in enumerate(mixed):
        z, _, _, _ = oracle.solve(prob, 1)
        ref = z[: prob.nvars]
        if strict:
            assert np.array_equal(sol[p], ref)
        else:
            assert np.linalg.norm(sol[p] - ref) / np.linalg.norm(ref) <= REL_TOL
            ok, detail = _kkt_ok(oracle, prob, sol[p])
            assert ok, detail
    # a second new right-hand side against the same cached factorisation (back to the first one)
    bs.set_rhs_flat(*[np.stack([getattr(p, k) for p in first]) for k in ("q", "r", "d", "x0")])
    assert bs.solve_rhs_only() == 0
    sol = bs.solutions()
    for p, prob in enumerate(first):
        ref = oracle.solve(prob, 1)[0][: prob.nvars]
        if strict:
            assert np.array_equal(sol[p], ref)
        else:
            assert np.linalg.norm(sol[p] - ref) / np.linalg.norm(ref) <= REL_TOL
    bs.close()
    # new matrices invalidate the cached factorisation
    bs2 = ndlqr.BatchSolver(n, m, N, batch, flags=fl)
    bs2.initialize_flat(*stack(first))
    assert bs2.solve() == 0
    bs2.initialize_flat(*stack(other))
    assert bs2.solve_rhs_only() == -1
    bs2.close()
    # without a cached factorisation the call is refused
    bs = ndlqr.BatchSolver(n, m, N, batch)
    bs.initialize_flat(*stack(first))
    assert bs.solve() == 0
    if (n, m) in ((12, 4), (6, 3)):
        assert bs.solve_rhs_only() == -1
    bs.close()


# ---------------------------------------------------------------------------------------------
# BASELINE.json's full sizes: the oracle is too slow to check every member, so parity rests on
# size-independent properties -- the KKT residual of the raw problem (SURVEY.md 8c secondary
# witness), linearity of the solution in the right-hand side, determinism -- plus a full oracle
# comparison of a few sampled members.
@pytest.mark.parametrize("n,m,N,batch,sample", [(12, 4, 256, 1024, 12), (12, 4, 1024, 512, 4), (6, 3, 256, 1, 1),
                                                (64, 16, 512, 4, 1), (64, 16, 512, 256, 1)])
def test_full_size_properties(ndlqr, oracle, n, m, N, batch, sample):
    bs = ndlqr.BatchSolver(n, m, N, batch)
    bs.initialize_synthetic(1)
    assert bs.solve() == 0 and bs.cholesky_failures() == 0
    sol = bs.solutions()
    assert np.isfinite(sol).all()
    res, bn = bs.kkt_residuals()  # every member, on the device
    assert (res <= 1e-9 * np.maximum(1.0, bn)).all(), float((res / np.maximum(1.0, bn)).max())
    rng = np.random.default_rng(n * 1000 + N)
    picks = sorted(set([0, batch - 1] + list(rng.integers(0, batch, size=sample))))
    for p in picks:
        prob = synth(ndlqr, n, m, N, 1 + p)  # problem p of ndlqr_InitializeBatchSynthetic(seed0 = 1)
        ok, info = _kkt_ok(oracle, prob, sol[p])
        assert ok, (p, info)
    # full oracle comparison on two members (one at the (64,16,512) sizes: 340 MB of factor array
    # per oracle solve); threads only speed the oracle up
    for p in picks[:1 if n >= 64 else 2]:
        prob = synth(ndlqr, n, m, N, 1 + p)
        z, _, _, _ = oracle.solve(prob, 8)
        ref = z[: prob.nvars]
        assert np.linalg.norm(sol[p] - ref) / np.linalg.norm(ref) <= REL_TOL
    # determinism: the same launch sequence gives the same bits
    assert bs.solve() == 0
    assert np.array_equal(bs.solutions(), sol)
    bs.close()


@pytest.mark.parametrize("n,m,N", [(12, 4, 64), (6, 3, 16), (5, 2, 8), (16, 8, 4)])
def test_device_kkt_residual_matches_oracle(ndlqr, oracle, n, m, N):
    """ndlqr_BatchKktResiduals evaluates the same rows as the oracle's KKT residual: tiny for the
    solution, and -- with the right-hand side swapped under a stale solution -- the same large
    number the oracle computes for that (solution, problem) pair."""
    batch = 3
    first = [synth(ndlqr, n, m, N, 40 + p) for p in range(batch)]
    other = [synth(ndlqr, n, m, N, 90 + p) for p in range(batch)]
    mixed = [Problem(n, m, N, f.A, f.B, f.Q, f.R, o.q, o.r, o.d, o.x0) for f, o in zip(first, other)]
    bs = ndlqr.BatchSolver(n, m, N, batch)
    bs.initialize_flat(*stack(first))
    assert bs.solve() == 0
    sol = bs.solutions()
    res, bn = bs.kkt_residuals()
    for p, prob in enumerate(first):
        ores, obn = oracle.kkt_residual(prob, sol[p])
        assert abs(bn[p] - obn) <= 1e-12 * obn
        assert res[p] <= 1e-9 * max(1.0, bn[p]) and ores <= 1e-9 * max(1.0, obn)
    bs.set_rhs_flat(*[np.stack([getattr(p, k) for p in mixed]) for k in ("q", "r", "d", "x0")])
    res, bn = bs.kkt_residuals()  # stale solution against the new right-hand side
    for p, prob in enumerate(mixed):
        ores, obn = oracle.kkt_residual(prob, sol[p])
        assert ores > 1e-3
        assert abs(res[p] - ores) <= 1e-10 * ores, (res[p], ores)
        assert abs(bn[p] - obn) <= 1e-12 * obn
    bs.close()


def test_linearity_in_the_right_hand_side(ndlqr):
    """x(rhs1) + x(rhs2) == x(rhs1 + rhs2) for fixed A, B, Q, R (the KKT system is linear), checked
    at the headline size on every member."""
    n, m, N, batch = 12, 4, 256, 256
    base = [ndlqr.generate_synthetic(n, m, N, 10 + p) for p in range(batch)]
    alt = [ndlqr.generate_synthetic(n, m, N, 5000 + p) for p in range(batch)]
    st = lambda lst, k: np.stack([g[k] for g in lst])
    mats = [st(base, k) for k in ("A", "B", "Q", "R")]
    r1 = [st(base, k) for k in ("q", "r", "d", "x0")]
    r2 = [st(alt, k) for k in ("q", "r", "d", "x0")]
    rsum = [a + b for a, b in zip(r1, r2)]
    out = []
    for rhs in (r1, r2, rsum):
        bs = ndlqr.BatchSolver(n, m, N, batch)
        bs.initialize_flat(*mats, *rhs)
        assert bs.solve() == 0
        out.append(bs.solutions())
        bs.close()
    err = np.linalg.norm(out[0] + out[1] - out[2], axis=1) / np.linalg.norm(out[2], axis=1)
    assert err.max() <= 1e-10, err.max()


@pytest.mark.parametrize("n,m,N", [(12, 4, 2), (12, 4, 4), (6, 3, 2), (6, 3, 4), (13, 4, 8), (4, 2, 8)])
def test_short_horizons_specialised_shapes(ndlqr, oracle, n, m, N):
    """Horizons too short for the fused kernels fall back to shorter fusion / the generic path."""
    probs = [synth(ndlqr, n, m, N, 40 + p) for p in range(4)]
    bs = ndlqr.BatchSolver(n, m, N, 4, flags=ndlqr.FLAG_STRICT_FP | ndlqr.FLAG_KEEP_FACT)
    bs.initialize_flat(*stack(probs))
    assert bs.solve() == 0
    for p, prob in enumerate(probs):
        z, fact, _, _ = oracle.solve(prob, 1, want_fact=True)
        assert np.array_equal(bs.solution(p), z[: prob.nvars])
        assert np.array_equal(bs.factors(p), fact)
    bs.close()
    bs = ndlqr.BatchSolver(n, m, N, 4)  # fast mode, solution only
    bs.initialize_flat(*stack(probs))
    assert bs.solve() == 0
    for p, prob in enumerate(probs):
        ref = oracle.solve(prob, 1)[0][: prob.nvars]
        assert np.linalg.norm(bs.solution(p) - ref) / np.linalg.norm(ref) <= REL_TOL
    bs.close()


def test_device_side_packing_matches_host_packing(ndlqr):
    """ndlqr_InitializeBatchFlatDevice (pack kernel, inputs already in HBM -- device memory comes
    from torch here) gives the same bits as the host-packed upload. Runs in a fresh process: torch
    ships its own HIP runtime next to the system one this library links, and which of the two may
    still open the device depends on what the test process did before."""
    import subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = """
import sys
sys.path.insert(0, %r)
import numpy as np, torch
import rslqr_amd as R
n, m, N, batch = 12, 4, 64, 6
gen = [R.generate_synthetic(n, m, N, 60 + p) for p in range(batch)]
flat = [np.stack([g[k] for g in gen]) for k in ("A", "B", "Q", "R", "q", "r", "d", "x0")]
tens = [torch.from_numpy(np.ascontiguousarray(a)).to("cuda:0") for a in flat]
torch.cuda.synchronize()
host = R.BatchSolver(n, m, N, batch, device=0)
host.initialize_flat(*flat)
assert host.solve() == 0
dev = R.BatchSolver(n, m, N, batch, device=0)
dev.initialize_flat_device(*[t.data_ptr() for t in tens])
assert dev.solve() == 0
assert np.array_equal(dev.solutions(), host.solutions())
res, bn = dev.kkt_residuals()
assert (res <= 1e-9 * np.maximum(1.0, bn)).all()
# new matrices packed ON the device invalidate cached records / factors, like the host upload does
# (mirror of test_factor_solve_split's host-side check): the rhs-only re-solve must be refused
for fl in (R.FLAG_KEEP_RECORDS, R.FLAG_KEEP_FACT):
    keep = R.BatchSolver(n, m, N, batch, device=0, flags=fl)
    keep.initialize_flat_device(*[t.data_ptr() for t in tens])
    assert keep.solve() == 0
    assert keep.solve_rhs_only() == 0
    gen2 = [R.generate_synthetic(n, m, N, 160 + p) for p in range(batch)]
    tens2 = [torch.from_numpy(np.ascontiguousarray(np.stack([g[k] for g in gen2]))).to("cuda:0")
             for k in ("A", "B", "Q", "R", "q", "r", "d", "x0")]
    torch.cuda.synchronize()
    keep.initialize_flat_device(*[t.data_ptr() for t in tens2])
    assert keep.solve_rhs_only() == -1, "stale records accepted after a device-side re-pack"
    assert keep.solve() == 0 and keep.solve_rhs_only() == 0
    keep.close()
# packed solutions written to device memory (the send buffer of the multi-GPU gather)
out = torch.empty((batch, dev.nvars), dtype=torch.float64, device="cuda:0")
dev.solutions_to_device(out.data_ptr())
dev.synchronize()
assert np.array_equal(out.cpu().numpy(), dev.solutions())
print("DEVICE_PACKING_OK")
""" % root
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "DEVICE_PACKING_OK" in out.stdout, out.stdout[-2000:] + out.stderr[-3000:]


@pytest.mark.parametrize("n,m,N,batch,flags", [(12, 4, 64, 300, 0), (6, 3, 32, 5, 0), (20, 6, 16, 40, 0),
                                                 (12, 4, 64, 300, 16), (7, 2, 8, 3, 0)])
def test_step_async_transfers_and_solves(ndlqr, oracle, n, m, N, batch, flags):
    """ndlqr_BatchStepAsync (the MPC step: new q, r, d, x0 up, factor + solve, solutions down; what the reference
    does per iteration with Reset + Initialize + Solve + CopySolution, src/solve.h:20-32): a stream of steps with a
    different right-hand side each, two in flight, pinned host arrays; every step's solutions against the oracle's
    solve of that problem; the pageable-memory form; the plain download functions afterwards."""
    gens = [ndlqr.generate_synthetic(n, m, N, 500 + p) for p in range(batch)]
    flat = {k: np.stack([g[k] for g in gens]) for k in ("A", "B", "Q", "R", "q", "r", "d", "x0")}
    bs = ndlqr.BatchSolver(n, m, N, batch, flags=flags)
    bs.initialize_flat(*[flat[k] for k in ("A", "B", "Q", "R", "q", "r", "d", "x0")])
    rng = np.random.default_rng(7)
    nsteps = 5
    ins, outs = [], []
    for s in range(nsteps):
        arrs = {}
        for k in ("q", "r", "d", "x0"):
            a = ndlqr.pinned_empty(flat[k].shape)
            a[...] = flat[k] + 0.25 * (s + 1) * rng.standard_normal(flat[k].shape)
            arrs[k] = a
        ins.append(arrs)
        outs.append(ndlqr.pinned_empty((batch, bs.nvars)))
        outs[-1][...] = np.nan
    for s in range(nsteps):
        assert bs.step_async(ins[s]["q"], ins[s]["r"], ins[s]["d"], ins[s]["x0"], outs[s]) == 0
        if s >= 1:
            assert bs.synchronize_previous() == 0
            assert np.isfinite(outs[s - 1]).all()  # the older step is complete while the newer one is in flight
    assert bs.synchronize() == 0
    assert bs.cholesky_failures() == 0
    for s in range(nsteps):
        for p in sorted({0, batch // 2, batch - 1}):
            prob = Problem(n, m, N, flat["A"][p], flat["B"][p], flat["Q"][p], flat["R"][p], ins[s]["q"][p],
                           ins[s]["r"][p], ins[s]["d"][p], ins[s]["x0"][p])
            ref = oracle.solve(prob, 1)[0][: prob.nvars]
            assert np.linalg.norm(outs[s][p] - ref) / np.linalg.norm(ref) <= REL_TOL, (s, p)
    # the solution resident on the device is the last step's: both download paths (pageable: staged, pinned: direct)
    last = outs[-1].copy()
    assert np.array_equal(bs.solutions(), last)
    pinned = ndlqr.pinned_empty((batch, bs.nvars))
    assert np.array_equal(bs.solutions(out=pinned), last)
    assert np.array_equal(bs.solution(batch - 1), last[batch - 1])
    res, bn = bs.kkt_residuals()  # against the right-hand side of the step it belongs to
    assert (res <= 1e-9 * np.maximum(1.0, bn)).all()
    # pageable arrays: the call blocks but is correct
    out = np.zeros((batch, bs.nvars))
    assert bs.step_async(ins[0]["q"].copy(), ins[0]["r"].copy(), ins[0]["d"].copy(), ins[0]["x0"].copy(), out) == 0
    assert bs.synchronize() == 0
    if flags & ndlqr.FLAG_KEEP_RECORDS:
        # (step 0 was the factor + solve that left the records; this one is the right-hand-side re-solve on them: the
        #  same numbers to rounding, by another sequence of operations)
        assert np.linalg.norm(out - outs[0]) / np.linalg.norm(outs[0]) <= 1e-12
        assert bs.schedule().startswith("reduced-compact-records")
    else:
        assert np.array_equal(out, outs[0])
    # a new right-hand side for every following solve replaces both buffer sets' copies
    bs.set_rhs_flat(flat["q"], flat["r"], flat["d"], flat["x0"])
    assert bs.solve_async() == 0 and bs.solve_async() == 0 and bs.synchronize() == 0
    prob = Problem(n, m, N, *[flat[k][0] for k in ("A", "B", "Q", "R", "q", "r", "d", "x0")])
    ref = oracle.solve(prob, 1)[0][: prob.nvars]
    assert np.linalg.norm(bs.solution(0) - ref) / np.linalg.norm(ref) <= REL_TOL
    # MPC steps that replace x0 alone (q, r, d: what set_rhs_flat left in both buffer sets), two in flight
    xs = [ndlqr.pinned_empty(flat["x0"].shape) for _ in range(3)]
    for s, x in enumerate(xs):
        x[...] = flat["x0"] + (s + 1.0)
        assert bs.step_async(None, None, None, x, outs[s]) == 0
    assert bs.synchronize() == 0
    for s, x in enumerate(xs):
        p = batch - 1
        prob = Problem(n, m, N, flat["A"][p], flat["B"][p], flat["Q"][p], flat["R"][p], flat["q"][p], flat["r"][p],
                       flat["d"][p], x[p])
        ref = oracle.solve(prob, 1)[0][: prob.nvars]
        assert np.linalg.norm(outs[s][p] - ref) / np.linalg.norm(ref) <= REL_TOL, s
    bs.close()


@pytest.mark.parametrize("n,m,N,batch", [(12, 4, 64, 300), (6, 3, 32, 5), (7, 9, 16, 3), (20, 6, 16, 40)])
def test_step_flows_mix_freely(ndlqr, oracle, n, m, N, batch):
    """There is ONE logical right-hand side; the two buffer sets of the pipeline each hold a copy and a step writes its
    own set only. Full steps with a different q, r, d each, then x0-only steps, then a plain solve: every result is the
    oracle's solve with the most recently written parts (round 3 solved set B's x0-only step against the q, r, d of
    the full step before the last)."""
    gens = [ndlqr.generate_synthetic(n, m, N, 900 + p) for p in range(batch)]
    flat = {k: np.stack([g[k] for g in gens]) for k in ("A", "B", "Q", "R", "q", "r", "d", "x0")}
    bs = ndlqr.BatchSolver(n, m, N, batch)
    bs.initialize_flat(*[flat[k] for k in ("A", "B", "Q", "R", "q", "r", "d", "x0")])
    rng = np.random.default_rng(3)
    pin = lambda a: (lambda b: (b.__setitem__(Ellipsis, a), b)[1])(ndlqr.pinned_empty(a.shape))
    outs = [ndlqr.pinned_empty((batch, bs.nvars)) for _ in range(8)]
    cur = {k: flat[k] for k in ("q", "r", "d", "x0")}
    want = []

    def check(out, parts, label):
        for p in sorted({0, batch - 1}):
            prob = Problem(n, m, N, flat["A"][p], flat["B"][p], flat["Q"][p], flat["R"][p], parts["q"][p],
                           parts["r"][p], parts["d"][p], parts["x0"][p])
            ref = oracle.solve(prob, 1)[0][: prob.nvars]
            assert np.linalg.norm(out[p] - ref) / np.linalg.norm(ref) <= REL_TOL, (label, p)

    step = 0
    for s in range(3):  # three full steps (sets A, B, A), each with its own q, r, d, x0
        cur = {k: pin(flat[k] + 0.3 * (s + 1) * rng.standard_normal(flat[k].shape)) for k in ("q", "r", "d", "x0")}
        assert bs.step_async(cur["q"], cur["r"], cur["d"], cur["x0"], outs[step]) == 0
        want.append((outs[step], dict(cur), "full %d" % s)); step += 1
    for s in range(3):  # x0-only steps: set B first, whose q, r, d are those of the full step before the last
        cur = dict(cur); cur["x0"] = pin(flat["x0"] + 2.0 + s)
        assert bs.step_async(None, None, None, cur["x0"], outs[step]) == 0
        want.append((outs[step], dict(cur), "x0-only %d" % s)); step += 1
    cur = dict(cur); cur["d"] = pin(flat["d"] * 0.5); cur["x0"] = pin(flat["x0"] - 1.0)  # d and x0 alone
    assert bs.step_async(None, None, cur["d"], cur["x0"], outs[step]) == 0
    want.append((outs[step], dict(cur), "d + x0")); step += 1
    assert bs.synchronize() == 0
    for out, parts, label in want:
        check(out, parts, label)
    # plain solves behind the steps (either buffer set) see the latest right-hand side
    for _ in range(2):
        assert bs.solve() == 0
        check(bs.solutions(), cur, "plain solve")
        res, bn = bs.kkt_residuals()
        assert (res <= 1e-9 * np.maximum(1.0, bn)).all()
    # an upload of the right-hand side of SOME problems lands on the complete latest copy
    bs.close()


@pytest.mark.parametrize("n,m,N,batch", [(12, 4, 64, 40), (6, 3, 32, 5), (7, 9, 16, 3), (20, 6, 16, 4)])
def test_step_selection_and_slices(ndlqr, oracle, n, m, N, batch):
    """ndlqr_BatchSetStepSelection / ndlqr_CopyBatchSolutionSlices: the slice a step brings down is that slice of the
    full solution vector (src/solve.c:192-201 hands back all of it) -- u of knot 0, x and u of the first knots, every
    block of a knot range in the middle."""
    bs = ndlqr.BatchSolver(n, m, N, batch)
    bs.initialize_synthetic(77)
    assert bs.solve() == 0
    full = bs.solutions()
    zb = 2 * n + m
    padded = np.zeros((batch, N * zb)); padded[:, : bs.nvars] = full
    Z = padded.reshape(batch, N, zb)

    def expect(k0, nk, blocks):
        cols = ([*range(0, n)] if blocks & ndlqr.SOLN_LAMBDA else []) + ([*range(n, 2 * n)] if blocks & ndlqr.SOLN_STATE else []) + \
               ([*range(2 * n, zb)] if blocks & ndlqr.SOLN_INPUT else [])
        return Z[:, k0:k0 + nk, :][:, :, cols]

    cases = [(0, 1, ndlqr.SOLN_INPUT), (0, min(4, N), ndlqr.SOLN_STATE | ndlqr.SOLN_INPUT), (N // 2, N // 2, 7), (N - 1, 1, 3)]
    for k0, nk, blocks in cases:
        assert np.array_equal(bs.solution_slices(k0, nk, blocks), expect(k0, nk, blocks)), (k0, nk, blocks)
    with pytest.raises(ValueError):
        bs.set_step_selection(N - 1, 2, 7)
    with pytest.raises(ValueError):
        bs.set_step_selection(0, 1, 0)
    # steps that bring down the slice alone, two in flight, x0 replaced per step
    g = [ndlqr.generate_synthetic(n, m, N, 77 + p) for p in range(batch)]
    x0 = np.stack([gg["x0"] for gg in g])
    for k0, nk, blocks in cases[:2]:
        bs.set_step_selection(k0, nk, blocks)
        xs = [ndlqr.pinned_empty(x0.shape) for _ in range(3)]
        outs = [ndlqr.pinned_empty((batch, nk, bs.slice_width(blocks))) for _ in range(3)]
        for s in range(3):
            xs[s][...] = x0 * (1.0 + s)
            assert bs.step_async(None, None, None, xs[s], outs[s]) == 0
        assert bs.synchronize() == 0
        for s in range(3):
            p = batch - 1
            prob = Problem(n, m, N, g[p]["A"], g[p]["B"], g[p]["Q"], g[p]["R"], g[p]["q"], g[p]["r"], g[p]["d"], xs[s][p])
            ref = np.zeros(N * zb); ref[: prob.nvars] = oracle.solve(prob, 1)[0][: prob.nvars]
            cols = ([*range(n, 2 * n)] if blocks & 2 else []) + ([*range(2 * n, zb)] if blocks & 4 else [])
            want = ref.reshape(N, zb)[k0:k0 + nk][:, cols]
            assert np.linalg.norm(outs[s][p] - want) <= REL_TOL * max(1.0, np.linalg.norm(ref)), (s, blocks)
        assert np.array_equal(outs[2], bs.solution_slices(k0, nk, blocks))
    bs.set_step_selection()  # back to everything
    out = ndlqr.pinned_empty((batch, bs.nvars))
    x = ndlqr.pinned_empty(x0.shape); x[...] = x0
    assert bs.step_async(None, None, None, x, out) == 0 and bs.synchronize() == 0
    assert np.linalg.norm(out - full) <= 1e-12 * np.linalg.norm(full)
    bs.close()


@pytest.mark.parametrize("n,m,N,batch,flags,want", [(12, 4, 64, 200, 0, "reduced-fused2"), (13, 4, 64, 160, 0, "reduced"),
                                                     (6, 3, 64, 300, 0, "reduced"), (12, 4, 256, 40, 16, "reduced-compact-records"),
                                                     (11, 3, 128, 80, 0, "reduced-fused2"), (12, 4, 64, 3, 0, "reduced-tree"),
                                                     (7, 9, 16, 3, 0, None), (20, 6, 16, 4, 0, "generic-reduced"),
                                                     (32, 8, 64, 6, 0, "generic-reduced"), (50, 10, 32, 3, 0, "generic-reduced"),
                                                     (20, 6, 32, 4, 16, "generic-reduced-records"), (48, 16, 16, 2, 16, "generic-reduced-records (re-solve)"),
                                                     (5, 2, 2, 5, 0, None),
                                                     (13, 4, 4, 6, 0, None), (8, 4, 4, 3, 16, None), (36, 4, 2, 2, 0, None),
                                                     (4, 2, 64, 50, 0, "knot-lean"), (6, 3, 64, 3, 16, "reduced-tree"),
                                                     (12, 4, 1024, 2, 0, "reduced-tree")])
def test_step_computes_selection_alone(ndlqr, oracle, n, m, N, batch, flags, want):
    """NDLQR_SOLN_ONLY: a step whose caller wants nothing but a knot range runs the workgroups of the last launch of the
    back-substitution that hold it (the MPC step that computes u of knot 0: one workgroup per problem instead of N / 8)
    -- on the level-per-launch schedules, the re-solve on kept records, the runtime-sized separator-only schedule (of every
    level the separators above the range) the tree and knot-lean schedules (self-contained workgroups of backsub_small) and, computing everything, the others. The slices
    against the oracle with x0 replaced per step and steps in flight; afterwards the whole vector is refused until a
    step without the bit has run (the reference hands back all of it: src/solve.c:192-201)."""
    bs = ndlqr.BatchSolver(n, m, N, batch, flags=flags)
    bs.initialize_synthetic(91)
    g = [ndlqr.generate_synthetic(n, m, N, 91 + p) for p in range(batch)]
    x0 = np.stack([gg["x0"] for gg in g])
    zb = 2 * n + m
    ONLY = ndlqr.SOLN_ONLY
    ka = min(5, N - 1)  # (horizons below sixteen knots: the same cases, clipped)
    cases = [(0, 1, ndlqr.SOLN_INPUT), (ka, min(6, N - ka), ndlqr.SOLN_STATE | ndlqr.SOLN_INPUT), (max(N - 8, 0), min(8, N), 7),
             (N // 2 - 1, 2, 7)]
    for k0, nk, blocks in cases:
        bs.set_step_selection(k0, nk, blocks | ONLY)
        xs = [ndlqr.pinned_empty(x0.shape) for _ in range(4)]
        outs = [ndlqr.pinned_empty((batch, nk, bs.slice_width(blocks))) for _ in range(4)]
        for s in range(4):
            xs[s][...] = x0 * (1.0 - 0.5 * s)
            assert bs.step_async(None, None, None, xs[s], outs[s]) == 0
            if s >= 1:
                assert bs.synchronize_previous() == 0
        assert bs.synchronize() == 0
        if want is not None and k0 == 0 and nk == 1:
            assert bs.schedule().startswith(want), bs.schedule()
        for s in (0, 3):
            for p in (0, batch - 1):
                prob = Problem(n, m, N, g[p]["A"], g[p]["B"], g[p]["Q"], g[p]["R"], g[p]["q"], g[p]["r"], g[p]["d"], xs[s][p])
                ref = np.zeros(N * zb); ref[: prob.nvars] = oracle.solve(prob, 1)[0][: prob.nvars]
                cols = ([*range(0, n)] if blocks & 1 else []) + ([*range(n, 2 * n)] if blocks & 2 else []) + \
                       ([*range(2 * n, zb)] if blocks & 4 else [])
                wantv = ref.reshape(N, zb)[k0:k0 + nk][:, cols]
                assert np.linalg.norm(outs[s][p] - wantv) <= REL_TOL * max(1.0, np.linalg.norm(ref)), (k0, nk, s, p)
        # the slice is what the solver holds: inside it downloads work, the whole vector and other knots are refused
        assert np.array_equal(bs.solution_slices(k0, nk, blocks), outs[3])
        with pytest.raises(RuntimeError):
            bs.solutions()
        with pytest.raises(RuntimeError):
            bs.kkt_residuals()
        outside = [kk for kk in range(0, N, 8) if kk + 8 <= k0 - k0 % 8 or kk > k0 + nk - 1]
        if outside:
            with pytest.raises(RuntimeError):
                bs.solution_slices(outside[-1], 1, 7)
    # a step without the bit (the same selection) leaves the whole vector again
    bs.set_step_selection(0, 1, ndlqr.SOLN_INPUT)
    x = ndlqr.pinned_empty(x0.shape); x[...] = x0
    u0 = ndlqr.pinned_empty((batch, 1, m))
    assert bs.step_async(None, None, None, x, u0) == 0 and bs.synchronize() == 0
    full = bs.solutions()
    res, bn = bs.kkt_residuals()
    assert (res <= 1e-9 * np.maximum(1.0, bn)).all()
    assert np.array_equal(u0[:, 0, :], full[:, 2 * n:zb])
    # and a plain solve after an ONLY step as well
    bs.set_step_selection(0, 1, ndlqr.SOLN_INPUT | ONLY)
    assert bs.step_async(None, None, None, x, u0) == 0 and bs.synchronize() == 0
    bs.set_step_selection()
    assert bs.solve() == 0
    assert np.linalg.norm(bs.solutions() - full) <= 1e-12 * np.linalg.norm(full)
    bs.close()


@pytest.mark.parametrize("n,m,N,batch,flags", [(12, 4, 64, 200, 0), (6, 3, 32, 5, 0), (12, 4, 128, 80, 16), (7, 9, 16, 3, 0)])
def test_step_on_device_pointers(ndlqr, oracle, n, m, N, batch, flags):
    """ndlqr_BatchStepAsync with q, r, d, x0 and soln in the solver's device memory (ndlqr_DeviceAlloc): the pack kernels
    read and write them where they are -- the step of a loop that lives on the GPU. Same results as the step on pinned
    host arrays, bit for bit; whole vectors, a slice, and a mix of host and device pointers."""
    bs = ndlqr.BatchSolver(n, m, N, batch, flags=flags)
    bs.initialize_synthetic(57)
    g = [ndlqr.generate_synthetic(n, m, N, 57 + p) for p in range(batch)]
    q, r, d, x0 = (np.stack([gg[k] for gg in g]) for k in ("q", "r", "d", "x0"))
    hq, hr, hd, hx = (ndlqr.pinned_empty(a.shape) for a in (q, r, d, x0))
    hq[...], hr[...], hd[...], hx[...] = 0.5 * q, 2.0 * r, d, -x0
    host_out = ndlqr.pinned_empty((batch, bs.nvars))
    assert bs.step_async(hq, hr, hd, hx, host_out) == 0 and bs.synchronize() == 0
    dq, dr, dd, dx = (ndlqr.DeviceArray(a.shape).set(a) for a in (hq, hr, hd, hx))
    dev_out = ndlqr.DeviceArray((batch, bs.nvars))
    for _ in range(2):  # (both buffer sets)
        assert bs.step_async(dq, dr, dd, dx, dev_out) == 0
    assert bs.synchronize() == 0
    if flags == 0:
        assert np.array_equal(dev_out.get(), host_out)
    else:  # (records kept: the first step factored, the others are re-solves -- equal to rounding)
        assert np.linalg.norm(dev_out.get() - host_out) <= 1e-12 * np.linalg.norm(host_out)
    p = batch - 1
    prob = Problem(n, m, N, g[p]["A"], g[p]["B"], g[p]["Q"], g[p]["R"], hq[p], hr[p], hd[p], hx[p])
    ref = oracle.solve(prob, 1)[0][: prob.nvars]
    assert np.linalg.norm(host_out[p] - ref) <= REL_TOL * np.linalg.norm(ref)
    # x0 alone from the device, u of knot 0 into device memory, computed alone; then a mix: x0 on the device, slice on the host
    hx2 = -0.25 * x0
    dx.set(hx2)
    bs.set_step_selection(0, 1, ndlqr.SOLN_INPUT | ndlqr.SOLN_ONLY)
    du0 = ndlqr.DeviceArray((batch, 1, m))
    hu0 = ndlqr.pinned_empty((batch, 1, m))
    assert bs.step_async(None, None, None, dx, du0) == 0
    assert bs.step_async(None, None, None, dx, hu0) == 0
    assert bs.synchronize() == 0
    assert np.array_equal(du0.get(), hu0)
    prob = Problem(n, m, N, g[p]["A"], g[p]["B"], g[p]["Q"], g[p]["R"], hq[p], hr[p], hd[p], hx2[p])
    ref = oracle.solve(prob, 1)[0][: prob.nvars]
    assert np.linalg.norm(hu0[p, 0] - ref[2 * n:2 * n + m]) <= REL_TOL * np.linalg.norm(ref)
    bs.close()


@pytest.mark.parametrize("n,m,N,batch", [(12, 4, 64, 200), (13, 4, 64, 160), (6, 3, 32, 5), (20, 6, 32, 4), (7, 9, 16, 3),
                                         (21, 5, 16, 3), (33, 7, 16, 2), (64, 16, 16, 2)])
def test_solve_delivering_a_slice_alone(ndlqr, oracle, n, m, N, batch):
    """ndlqr_SolveBatchSlicesAsync: factor + solve computing and delivering a knot range alone -- the solve of a loop that
    replaces A, B, Q, R every iteration (here: device-side packing of new problems from device memory, ndlqr_DeviceAlloc)
    and consumes u of knot 0 (the pack kernel in its three forms: eight knots per workgroup, through LDS, direct). Slices
    against the oracle and bit for bit against a plain solve; two calls in flight; the
    whole vector is refused afterwards and back after a plain solve."""
    bs = ndlqr.BatchSolver(n, m, N, batch)
    zb = 2 * n + m
    keys = ("A", "B", "Q", "R", "q", "r", "d", "x0")
    for it, (k0, nk, blocks) in enumerate([(0, 1, ndlqr.SOLN_INPUT), (3, 7, ndlqr.SOLN_STATE | ndlqr.SOLN_INPUT), (N - 2, 2, 7)]):
        gens = [ndlqr.generate_synthetic(n, m, N, 700 + 50 * it + p) for p in range(batch)]
        flat = [np.stack([g[k] for g in gens]) for k in keys]
        dev = [ndlqr.DeviceArray(a.shape).set(a) for a in flat]
        bs.initialize_flat_device(*[a.ptr for a in dev])
        width = bs.slice_width(blocks)
        outs = [ndlqr.DeviceArray((batch, nk, width)), ndlqr.pinned_empty((batch, nk, width))]
        for o in outs:  # (two in flight: one per buffer set)
            assert bs.solve_slices_async(k0, nk, blocks, o) == 0
        assert bs.synchronize() == 0
        got = outs[1]
        assert np.array_equal(outs[0].get(), got)
        with pytest.raises(RuntimeError):
            bs.solutions()
        assert np.array_equal(bs.solution_slices(k0, nk, blocks), got)
        assert bs.solve() == 0
        full = bs.solutions()
        cols = ([*range(0, n)] if blocks & 1 else []) + ([*range(n, 2 * n)] if blocks & 2 else []) + ([*range(2 * n, zb)] if blocks & 4 else [])
        padded = np.zeros((batch, N * zb)); padded[:, : bs.nvars] = full
        assert np.array_equal(got, padded.reshape(batch, N, zb)[:, k0:k0 + nk, :][:, :, cols]), (k0, nk, blocks)
        for p in (0, batch - 1):
            g = gens[p]
            prob = Problem(n, m, N, g["A"], g["B"], g["Q"], g["R"], g["q"], g["r"], g["d"], g["x0"])
            ref = oracle.solve(prob, 1)[0][: prob.nvars]
            assert np.linalg.norm(full[p] - ref) <= REL_TOL * np.linalg.norm(ref)
    with pytest.raises(AssertionError):
        bs.solve_slices_async(0, 2, 7, ndlqr.pinned_empty((batch, 1, zb)))
    assert bs.L.ndlqr_SolveBatchSlicesAsync(bs.h, N - 1, 2, 7, None) != 0
    bs.close()


def test_large_download_through_bounce_buffers(ndlqr):
    """ndlqr_CopyBatchSolutions into pageable memory goes through two pinned 8 MB bounce buffers in chunks: a
    download larger than several chunks equals the pinned (single-copy) one and the per-problem one."""
    n, m, N, batch = 12, 4, 256, 600  # 600 x 7164 doubles = 34 MB: five chunks
    bs = ndlqr.BatchSolver(n, m, N, batch)
    bs.initialize_synthetic(77)
    assert bs.solve() == 0
    pageable = bs.solutions()
    pinned = bs.solutions(out=ndlqr.pinned_empty((batch, bs.nvars)))
    assert np.array_equal(pageable, pinned)
    for p in (0, 311, batch - 1):
        assert np.array_equal(bs.solution(p), pageable[p])
    res, bn = bs.kkt_residuals()
    assert (res <= 1e-9 * np.maximum(1.0, bn)).all()
    bs.close()


@pytest.mark.parametrize("n,m,N,batch", [(7, 9, 16, 3), (5, 3, 32, 2), (1, 1, 8, 2), (11, 3, 64, 2), (14, 2, 8, 2),
                                         (9, 6, 16, 40), (7, 9, 64, 300)])
def test_padded_shapes(ndlqr, oracle, n, m, N, batch, monkeypatch):
    """A block size without a size-specialised instance (below 16 states, N >= 8) runs zero-padded inside the
    cheapest instance that contains it: dummy states / inputs with unit weights and no coupling, set once on the
    device; only the boundary functions (uploads, device-side packing, downloads, factor download, MPC step) know
    the caller's block size. Every mode against the oracle on the caller's problem: fast (and against the
    runtime-sized kernels on the own block size), strict (bit-exact: the pad terms are exact zeros), KEEP_FACT
    factors, rhs-only re-solve on kept records, the MPC step, the device-side KKT residual, a non-SPD weight."""
    probs = [synth(ndlqr, n, m, N, 4200 + p) for p in range(batch)]
    sample = sorted({0, batch // 2, batch - 1})
    refs = {p: oracle.solve(probs[p], 1, want_fact=True) for p in sample}
    bs = ndlqr.BatchSolver(n, m, N, batch)
    bs.initialize_flat(*stack(probs))
    for _ in range(3):
        assert bs.solve() == 0
    assert bs.schedule() in ("reduced", "reduced-fused2", "reduced-tree")  # the instance's schedule, not the runtime-sized one
    fast = bs.solutions()
    for p in sample:
        ref = refs[p][0][: probs[p].nvars]
        assert np.linalg.norm(fast[p] - ref) / np.linalg.norm(ref) <= REL_TOL
    res, bn = bs.kkt_residuals()
    assert (res <= 1e-9 * np.maximum(1.0, bn)).all()
    for p in sample:  # the device-side residual is that of the caller's problem (the dummies contribute exact zeros)
        r_o, b_o = oracle.kkt_residual(probs[p], fast[p])
        assert abs(bn[p] - b_o) <= 1e-12 * b_o and res[p] <= 1e-9 * max(1.0, b_o)
    # strict mode: bit-identical solution AND factor array (un-padded by the download)
    bs.set_flags(ndlqr.FLAG_STRICT_FP | ndlqr.FLAG_KEEP_FACT)
    assert bs.solve() == 0
    for p in sample:
        z, fact, _, _ = refs[p]
        assert np.array_equal(bs.solution(p), z[: probs[p].nvars])
        assert np.array_equal(bs.factors(p), fact)
    # fast mode with the factor array kept
    bs.set_flags(ndlqr.FLAG_KEEP_FACT)
    assert bs.solve() == 0
    f = bs.factors(sample[-1])
    assert np.linalg.norm(f - refs[sample[-1]][1]) / np.linalg.norm(refs[sample[-1]][1]) <= REL_TOL
    # rhs-only re-solve on kept records, new right-hand side through the caller's layout
    bs.set_flags(ndlqr.FLAG_KEEP_RECORDS)
    assert bs.solve() == 0
    other = [synth(ndlqr, n, m, N, 8800 + p) for p in range(batch)]
    bs.set_rhs_flat(*[np.stack([getattr(o, k) for o in other]) for k in ("q", "r", "d", "x0")])
    assert bs.solve_rhs_only() == 0
    for p in sample:
        mixed = Problem(n, m, N, probs[p].A, probs[p].B, probs[p].Q, probs[p].R, other[p].q, other[p].r, other[p].d,
                        other[p].x0)
        ref = oracle.solve(mixed, 1)[0][: mixed.nvars]
        assert np.linalg.norm(bs.solution(p) - ref) / np.linalg.norm(ref) <= REL_TOL
    # the MPC step (pack kernel from the caller's flat layout into the padded right-hand side, packed solutions back)
    bs.set_flags(0)
    arrs = []
    for k in ("q", "r", "d", "x0"):
        a = np.stack([getattr(o, k) for o in probs])
        pa = ndlqr.pinned_empty(a.shape)
        pa[...] = a
        arrs.append(pa)
    out = ndlqr.pinned_empty((batch, bs.nvars))
    assert bs.step_async(arrs[0], arrs[1], arrs[2], arrs[3], out) == 0 and bs.synchronize() == 0
    assert np.array_equal(out, fast)
    ptrs = (C.c_void_p * 5)()  # no raw device pointers: the arrays have the padded layout
    assert ndlqr.lib().ndlqr_hip_device_pointers(bs.ctx, ptrs) == -1
    bs.close()
    # the same block size on the runtime-sized kernels: same answer to rounding
    monkeypatch.setenv("NDLQR_NO_PAD", "1")
    own = ndlqr.BatchSolver(n, m, N, batch)
    own.initialize_flat(*stack(probs))
    assert own.solve() == 0 and own.schedule().startswith("generic")
    assert np.linalg.norm(own.solutions() - fast) / np.linalg.norm(fast) <= REL_TOL
    own.close()
    monkeypatch.delenv("NDLQR_NO_PAD")
    # a non-positive weight is reported
    bad = probs[0]
    R = bad.R.copy()
    R[N // 2, 0] = -1.0
    one = ndlqr.BatchSolver(n, m, N, 1)
    one.initialize_flat(*[np.asarray(a)[None] for a in (bad.A, bad.B, bad.Q, R, bad.q, bad.r, bad.d, bad.x0)])
    assert one.solve() == -3 and one.cholesky_failures() >= 1
    one.close()


@pytest.mark.parametrize("n,m", [(12, 4), (13, 4), (15, 2), (12, 8), (8, 16), (9, 3), (10, 4), (6, 3), (8, 4), (14, 2),
                                 (11, 3), (7, 9)])
def test_level_per_launch_schedule_every_instance(ndlqr, oracle, monkeypatch, n, m):
    """The level-per-launch separator-only schedule (compact level-0 records: the panel rides through the paired
    Cholesky pass of the matrix-core bottom kernel -- row-broadcast kernel up to 8 states --, L or S-bar^-1 in the
    record, the DPP substitution of rb_backsub, the last three levels + top-down sweep in one launch) at a batch
    that selects it, for every size-specialised instance incl. the buckets and three padded block sizes: three
    members against the oracle, every member's KKT residual on the device."""
    N, batch = 64, 160  # 160 x 16 = 2560 bottom wavefronts: beyond the tree schedule's range
    monkeypatch.setenv("NDLQR_FUSE2", "0")  # (read when the solver is created; the (12,4) instance defaults to fused2)
    bs = ndlqr.BatchSolver(n, m, N, batch)
    bs.initialize_synthetic(31000)
    for _ in range(3):
        assert bs.solve() == 0
    assert bs.schedule() == "reduced"
    sol = bs.solutions()
    if (n, m) in ((12, 4), (11, 3), (13, 4)):  # the default choice: level 2 inside the bottom launch at (12,4) only
        monkeypatch.delenv("NDLQR_FUSE2")
        bd = ndlqr.BatchSolver(n, m, N, batch)
        bd.initialize_synthetic(31000)
        for _ in range(2):
            assert bd.solve() == 0
        assert bd.schedule() == ("reduced" if n == 13 else "reduced-fused2")
        assert np.abs(bd.solutions() - sol).max() <= 1e-9 * np.abs(sol).max()
        bd.close()
    res, bn = bs.kkt_residuals()
    assert (res <= 1e-9 * np.maximum(1.0, bn)).all()
    for p in (0, 77, batch - 1):
        prob = synth(ndlqr, n, m, N, 31000 + p)
        ref = oracle.solve(prob, 1)[0][: prob.nvars]
        assert np.linalg.norm(sol[p] - ref) / np.linalg.norm(ref) <= REL_TOL, (n, m, p)
    bs.close()

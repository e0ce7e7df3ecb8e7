"""GPU: the reference's unit tests for the stage functions and dense helpers, replayed through the
drop-in C API (device-backed Matrix* helpers + ndlqr_* stage functions on the host mirrors):
test/linalg_test.c, test/linalg_custom_test.c, test/nested_dissection_test.c:11-237.
Dense helpers keep the reference's operation order (no FMA), so against the oracle they are
compared bit-exactly."""
import ctypes as C
import os

import numpy as np
import pytest

from support import GOLDEN, load_json_matrix, load_json_problem, split_block

pytestmark = pytest.mark.gpu
SAMPLE = os.path.join(GOLDEN, "sample_problem.json")
LQRPROB = os.path.join(GOLDEN, "lqr_prob.json").encode()
TOL = 1e-6


def mat(ndlqr, a):
    """column-major Matrix view over a fresh numpy buffer; returns (Matrix, buffer)."""
    a = np.asarray(a, dtype=np.float64)
    buf = np.ascontiguousarray(a.T).ravel().copy()
    rows, cols = (a.shape[0], 1) if a.ndim == 1 else a.shape
    return ndlqr.Matrix(rows, cols, buf.ctypes.data_as(C.POINTER(C.c_double))), buf


def test_matmul(ndlqr):  # test/linalg_test.c:6-41
    L = ndlqr.lib()
    A, _ = mat(ndlqr, np.full((3, 4), 4.0))
    B, Bb = mat(ndlqr, np.full((4, 5), 3.0))
    Cm, Cb = mat(ndlqr, np.full((3, 5), 2.0))
    L.MatrixMultiply(C.byref(A), C.byref(B), C.byref(Cm), False, False, 1.0, 1.0)
    assert np.array_equal(Cb, np.full(15, 50.0))
    L.MatrixMultiply(C.byref(A), C.byref(Cm), C.byref(B), True, False, 1.0, -200.0)
    assert np.array_equal(Bb, np.zeros(20))
    x, _ = mat(ndlqr, np.array([1.0, 2, 3, 4]))
    b, bb = mat(ndlqr, np.zeros(3))
    L.MatrixMultiply(C.byref(A), C.byref(x), C.byref(b), False, False, 1.0, 0.0)
    assert np.array_equal(bb, [40.0, 40.0, 40.0])


def test_matmul_transposes_random_vs_numpy(ndlqr):
    L = ndlqr.lib()
    rng = np.random.default_rng(0)
    for tA in (False, True):
        for tB in (False, True):
            a = rng.standard_normal((5, 7) if not tA else (7, 5))
            b = rng.standard_normal((7, 4) if not tB else (4, 7))
            c0 = rng.standard_normal((5, 4))
            A, _ = mat(ndlqr, a); B, _ = mat(ndlqr, b); Cm, Cb = mat(ndlqr, c0)
            L.MatrixMultiply(C.byref(A), C.byref(B), C.byref(Cm), tA, tB, 0.7, -1.3)
            want = 0.7 * (a.T if tA else a) @ (b.T if tB else b) - 1.3 * c0
            assert np.allclose(Cb.reshape(4, 5).T, want, rtol=1e-13, atol=1e-13)


def test_symmetric_multiply_and_cholesky(ndlqr):  # test/linalg_test.c:43-122
    L = ndlqr.lib()
    Asym = np.array([[9.0, -3, -6], [-3, 17, -10], [-6, -10, 38]])
    A, Ab = mat(ndlqr, Asym)
    X, _ = mat(ndlqr, np.array([[3.0, 1], [3, 1], [2, 1]]))
    B, Bb = mat(ndlqr, np.zeros((3, 2)))
    L.MatrixSymmetricMultiply(C.byref(A), C.byref(X), C.byref(B), 1.0, 0.0)
    assert np.array_equal(Bb, [6.0, 22, 28, 0, 4, 22])
    D, Db = mat(ndlqr, 9.0 * np.eye(5))
    rhs, rb = mat(ndlqr, 9.0 * np.arange(1, 6))
    assert L.MatrixCholeskyFactorize(C.byref(D)) == 0
    assert np.array_equal(np.diag(Db.reshape(5, 5)), np.full(5, 3.0))
    assert L.MatrixCholeskySolve(C.byref(D), C.byref(rhs)) == 0
    assert np.array_equal(rb, np.arange(1.0, 6.0))
    b, bb = mat(ndlqr, np.array([6.0, 22, 28]))
    assert L.MatrixCholeskyFactorize(C.byref(A)) == 0
    assert np.allclose(np.tril(Ab.reshape(3, 3).T), np.linalg.cholesky(Asym), rtol=1e-15)
    L.MatrixCholeskySolve(C.byref(A), C.byref(b))
    assert np.linalg.norm(bb - [3.0, 3, 2]) < TOL
    bad, _ = mat(ndlqr, np.array([[1.0, 2], [2, 1]]))  # not positive definite
    assert L.MatrixCholeskyFactorize(C.byref(bad)) != 0


def gen_test_solver(L):
    prob = L.ndlqr_ReadLQRProblemJSONFile(LQRPROB)
    n = prob.contents.lqrdata[0].contents.nstates
    m = prob.contents.lqrdata[0].contents.ninputs
    solver = L.ndlqr_NewNdLqrSolver(n, m, prob.contents.nhorizon)
    assert L.ndlqr_InitializeWithLQRProblem(prob, solver) == 0
    L.ndlqr_FreeLQRProblem(prob)
    return solver


def test_stage_functions_replay_reference_tests(ndlqr, oracle):
    L = ndlqr.lib()
    py, _ = load_json_problem(LQRPROB.decode())
    solver = gen_test_solver(L)
    s = solver.contents
    o = oracle.solver(py)
    OL = oracle.L
    fact = lambda: s.fact.contents.numpy()
    soln = lambda: s.soln.contents.numpy()
    # SolveLeaves (test/nested_dissection_test.c:11-114)
    assert L.ndlqr_SolveLeaves(solver) == 0
    for k in range(py.N):
        OL.oracle_solve_leaf(o.h, k)
    assert np.array_equal(fact(), o.fact()) and np.array_equal(soln(), o.soln())
    b = load_json_matrix(SAMPLE, "b").ravel()
    assert np.linalg.norm(soln()[: b.size] - b) < TOL
    # FactorInnerProduct (:116-136)
    assert L.ndlqr_FactorInnerProduct(s.data, s.fact, 0, 0, 0) == 0
    OL.oracle_inner_product(o.h, 0, 0, 0, 0)
    S, _, _ = split_block(fact(), py, 1, 0)
    assert abs(S[0, 0] - 1.0025) < TOL and abs(S[3, 0] - 0.05) < TOL and abs(S[3, 3] - 2.0) < TOL
    assert np.array_equal(fact(), o.fact())
    # ShurCompliment (:138-237)
    f = C.POINTER(ndlqr.NdFactor)()
    L.ndlqr_GetNdFactor(s.fact, 1, 0, C.byref(f))
    assert L.MatrixCholeskyFactorize(C.byref(f.contents.lambda_)) == 0
    OL.oracle_factor_separator(o.h, 0, 0)
    info = C.POINTER(ndlqr.api.CholeskyInfo)()
    L.ndlqr_GetSFactorization(s.cholfacts, 0, 0, C.byref(info))
    for upper in (1, 2):
        L.ndlqr_FactorInnerProduct(s.data, s.fact, 0, 0, upper)
        assert L.ndlqr_SolveCholeskyFactor(s.fact, info, 0, 0, upper) == 0
        assert L.ndlqr_ComputeShurCompliment(solver, 0, 0, upper) == 0
        OL.oracle_inner_product(o.h, 0, 0, 0, upper)
        OL.oracle_solve_chol_factor(o.h, 0, 0, upper)
        OL.oracle_compute_schur_compliment(o.h, 0, 0, upper)
        assert np.array_equal(fact(), o.fact())
        for i in range(2):
            lam, st, inp = split_block(fact(), py, i, upper)
            for blk, tag in ((lam, "y"), (st, "x"), (inp, "u")):
                assert np.linalg.norm(blk - load_json_matrix(SAMPLE, "E%d%d%s" % (i, upper, tag))) < TOL
    L.ndlqr_FreeNdLqrSolver(solver)


def test_c_example_compiles_and_runs(ndlqr, tmp_path):
    """examples/solve_json.c = the reference's canonical caller, compiled with plain gcc against
    include/ndlqr.h and linked to librslqr_amd.so (drop-in at the source level)."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "solve_json")
    libdir = os.path.dirname(ndlqr.library_path())
    subprocess.run(["gcc", "-std=c11", "-Wall", "-Werror", "-I" + os.path.join(root, "include"),
                    os.path.join(root, "examples", "solve_json.c"), "-L" + libdir, "-lrslqr_amd",
                    "-Wl,-rpath," + libdir, "-lm", "-o", exe], check=True)
    for fname, tol in (("lqr_prob.json", 1e-8), ("lqr_prob_256.json", 1e-6)):
        out = subprocess.run([exe, os.path.join(GOLDEN, fname)], check=True, capture_output=True, text=True).stdout
        line = [l for l in out.splitlines() if "||x - soln||_2" in l][0]
        assert float(line.split("=")[1].split()[0]) < tol, out


@pytest.mark.gpu
@pytest.mark.parametrize("args", [[], ["6", "3", "32", "5", "3"], ["7", "2", "16", "4", "2"]])
def test_mpc_batch_example(ndlqr, tmp_path, args):
    """examples/mpc_batch.c: factor once, re-solve for new right-hand sides, device-side KKT check
    of every problem after every solve (specialised and generic shapes)."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "mpc_batch")
    libdir = os.path.dirname(ndlqr.library_path())
    subprocess.run(["gcc", "-std=c11", "-Wall", "-Werror", "-I" + os.path.join(root, "include"),
                    os.path.join(root, "examples", "mpc_batch.c"), "-L" + libdir, "-lrslqr_amd",
                    "-Wl,-rpath," + libdir, "-lm", "-o", exe], check=True)
    out = subprocess.run([exe] + args, capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr
    assert out.stdout.count("re-solve") >= 2



@pytest.mark.gpu
@pytest.mark.parametrize("args", [[], ["6", "3", "32", "7", "5"], ["7", "9", "16", "3", "4"], ["20", "6", "16", "4", "3"],
                                  ["12", "4", "64", "300", "6", "1"], ["6", "3", "32", "7", "5", "1"],
                                  ["12", "4", "64", "300", "6", "0", "1"], ["12", "4", "64", "300", "6", "1", "1"],
                                  ["6", "3", "32", "7", "5", "0", "1"], ["8", "4", "128", "100", "5", "1", "1"]])
def test_mpc_step_example(ndlqr, tmp_path, args):
    """examples/mpc_step.c: the asynchronous MPC step in plain C -- x0 up, factor + solve, u of knot 0 down
    (ndlqr_BatchSetStepSelection), two steps in flight on pinned host memory; the program checks the KKT residual of
    every problem on the device and the slice against the resident solution (specialised, padded and generic shapes;
    sixth argument: records kept, seventh: NDLQR_SOLN_ONLY -- the steps compute u of knot 0 alone)."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "mpc_step")
    libdir = os.path.dirname(ndlqr.library_path())
    subprocess.run(["gcc", "-std=c11", "-Wall", "-Werror", "-I" + os.path.join(root, "include"),
                    os.path.join(root, "examples", "mpc_step.c"), "-L" + libdir, "-lrslqr_amd",
                    "-Wl,-rpath," + libdir, "-lm", "-o", exe], check=True)
    out = subprocess.run([exe] + args, capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "matches the resident solution" in out.stdout

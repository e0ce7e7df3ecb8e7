"""Shared test helpers: fixture loading, ctypes bindings for the oracle / reference checkers.

Only tests/, bench.py's cpu_baseline leg and __graft_entry__.smoke() may import this module's
Oracle / Reference classes (they are the *checkers*, never the product path).
"""
import ctypes as C
import json
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")
ORACLE_DIR = os.path.join(ROOT, "oracle")
ORACLE_SO = os.environ.get("NDLQR_ORACLE_LIBRARY") or os.path.join(ORACLE_DIR, "liboracle.so")  # (the sanitizer run points at its own build)
REF_SO = os.path.join(ORACLE_DIR, "_ref", "libref.so")

dp = C.POINTER(C.c_double)


def _p(a):
    return a.ctypes.data_as(dp)


def build_oracle():
    """(Re)build oracle/liboracle.so (and oracle/_ref/libref.so when /root/reference exists)."""
    subprocess.run(["make", "-s", "-C", ORACLE_DIR], check=True)


class Problem:
    """Flat LQR problem arrays. A: [N, n*n] col-major, B: [N, n*m] col-major, Q,q,d: [N, n],
    R,r: [N, m], x0: [n]. (Every knot carries all fields, like the reference's LQRData.)"""

    def __init__(self, n, m, N, A, B, Q, R, q, r, d, x0):
        self.n, self.m, self.N = n, m, N
        f = lambda a, shape: np.ascontiguousarray(np.asarray(a, dtype=np.float64).reshape(shape))
        self.A = f(A, (N, n * n)); self.B = f(B, (N, n * m))
        self.Q = f(Q, (N, n)); self.R = f(R, (N, m))
        self.q = f(q, (N, n)); self.r = f(r, (N, m)); self.d = f(d, (N, n))
        self.x0 = f(x0, (n,))

    @property
    def K(self):
        return int(np.log2(self.N))

    @property
    def nvars(self):
        return (2 * self.n + self.m) * self.N - self.m

    def arrays(self):
        return (self.A, self.B, self.Q, self.R, self.q, self.r, self.d, self.x0)

    def cargs(self):
        return tuple(_p(a) for a in self.arrays())


def load_json_problem(path):
    """Reference JSON format (src/json_utils.c:136-259): 2-D arrays are arrays of columns,
    lqrdata[i]["index"] is 1-based. Returns (Problem, soln or None)."""
    with open(path) as fh:
        j = json.load(fh)
    N = j["nhorizon"]
    knots = [None] * N
    for kd in j["lqrdata"]:
        knots[kd["index"] - 1] = kd
    n, m = knots[0]["nstates"], knots[0]["ninputs"]
    col = lambda key: np.array([np.asarray(kd[key], dtype=np.float64).reshape(-1) for kd in knots])
    prob = Problem(n, m, N, col("A"), col("B"), col("Q"), col("R"), col("q"), col("r"), col("d"),
                   j["x0"])
    soln = np.asarray(j["soln"], dtype=np.float64).reshape(-1) if "soln" in j else None
    return prob, soln


def load_json_matrix(path, name):
    """ReadMatrixJSONFile equivalent (src/json_utils.c:311-348): returns column-major flat data
    as a (rows, cols) numpy array."""
    with open(path) as fh:
        j = json.load(fh)
    a = np.asarray(j[name], dtype=np.float64)
    if a.ndim == 1:
        return a.reshape(-1, 1)
    return a.T.copy()  # array of columns -> (rows, cols)


class Oracle:
    """ctypes binding of oracle/liboracle.so (our CPU restatement)."""

    def __init__(self):
        if not os.path.exists(ORACLE_SO):
            build_oracle()
        L = self.L = C.CDLL(ORACLE_SO)
        L.oracle_new.restype = C.c_void_p
        L.oracle_new.argtypes = [C.c_int] * 3
        L.oracle_free.argtypes = [C.c_void_p]
        L.oracle_reset.argtypes = [C.c_void_p]
        for nm in ("oracle_data", "oracle_fact", "oracle_soln", "oracle_diag"):
            getattr(L, nm).restype = dp
            getattr(L, nm).argtypes = [C.c_void_p]
        L.oracle_initialize.argtypes = [C.c_void_p] + [dp] * 8
        L.oracle_solve_leaf.argtypes = [C.c_void_p, C.c_int]
        L.oracle_inner_product.argtypes = [C.c_void_p] + [C.c_int] * 4
        L.oracle_factor_separator.argtypes = [C.c_void_p, C.c_int, C.c_int]
        L.oracle_solve_chol_factor.argtypes = [C.c_void_p] + [C.c_int] * 3
        L.oracle_solve_chol_rhs.argtypes = [C.c_void_p] + [C.c_int] * 2
        L.oracle_update_schur.argtypes = [C.c_void_p] + [C.c_int] * 6
        L.oracle_compute_schur_compliment.argtypes = [C.c_void_p] + [C.c_int] * 3
        L.oracle_solve.argtypes = [C.c_void_p, C.c_int]
        L.oracle_solve_ex.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int]
        L.oracle_chol_failures.argtypes = [C.c_void_p]
        L.oracle_solve_flat.argtypes = [C.c_int] * 3 + [dp] * 8 + [C.c_int, dp, dp, dp]
        L.oracle_bench.argtypes = [C.c_int] * 5 + [dp] * 8 + [C.c_int, C.c_int, dp]
        L.oracle_kkt_residual.restype = C.c_double
        L.oracle_kkt_residual.argtypes = [C.c_int] * 3 + [dp] * 9 + [dp]

    def solver(self, prob):
        return OracleSolver(self, prob)

    def solve(self, prob, nthreads=1, want_fact=False):
        n, m, N = prob.n, prob.m, prob.N
        zb = 2 * n + m
        soln = np.zeros(N * zb)
        fact = np.zeros(N * prob.K * zb * n) if want_fact else None
        ms = C.c_double(0)
        fails = self.L.oracle_solve_flat(n, m, N, *prob.cargs(), nthreads, _p(soln),
                                         _p(fact) if want_fact else None, C.byref(ms))
        return soln, fact, ms.value, fails

    def kkt_residual(self, prob, z):
        z = np.ascontiguousarray(z, dtype=np.float64)
        full = np.zeros((2 * prob.n + prob.m) * prob.N + 2 * prob.n + prob.m)
        full[: z.size] = z
        bn = C.c_double(0)
        res = self.L.oracle_kkt_residual(prob.n, prob.m, prob.N, *prob.cargs(), _p(full),
                                         C.byref(bn))
        return res, bn.value


class OracleSolver:
    def __init__(self, oracle, prob):
        self.o, self.L, self.prob = oracle, oracle.L, prob
        self.h = self.L.oracle_new(prob.n, prob.m, prob.N)
        assert self.h
        self.L.oracle_initialize(self.h, *prob.cargs())
        n, m, N, K = prob.n, prob.m, prob.N, prob.K
        self.zb, self.fb = 2 * n + m, (2 * n + m) * n
        self._nf, self._ns = N * K * self.fb, N * self.zb

    def __del__(self):
        if getattr(self, "h", None):
            self.L.oracle_free(self.h)
            self.h = None

    def fact(self):
        return np.ctypeslib.as_array(self.L.oracle_fact(self.h), (self._nf,))

    def data(self):
        return np.ctypeslib.as_array(self.L.oracle_data(self.h), (self._nf,))

    def soln(self):
        return np.ctypeslib.as_array(self.L.oracle_soln(self.h), (self._ns,))

    def fact_block(self, k, level):
        """Returns (lambda n x n, state n x n, input m x n) views, shaped (rows, cols)."""
        return split_block(self.fact(), self.prob, k, level)


def split_block(fact, prob, k, level):
    n, m, N = prob.n, prob.m, prob.N
    fb = (2 * n + m) * n
    b = fact[(k + N * level) * fb:(k + N * level + 1) * fb]
    lam = b[: n * n].reshape(n, n).T
    st = b[n * n: 2 * n * n].reshape(n, n).T
    inp = b[2 * n * n:].reshape(n, m).T
    return lam, st, inp


def have_reference():
    return os.path.exists(REF_SO)


class Reference:
    """ctypes binding of oracle/_ref/libref.so = the real reference hot path + ref_driver.c."""

    def __init__(self):
        L = self.L = C.CDLL(REF_SO)
        L.ref_new_solver.restype = C.c_void_p
        L.ref_new_solver.argtypes = [C.c_int] * 3 + [dp] * 8
        L.ref_reinit.argtypes = [C.c_void_p] + [C.c_int] * 3 + [dp] * 8
        L.ref_free_solver.argtypes = [C.c_void_p]
        L.ref_solve.argtypes = [C.c_void_p, C.c_int]
        L.ref_solve_time_ms.restype = C.c_double
        L.ref_solve_time_ms.argtypes = [C.c_void_p]
        for nm in ("ref_soln", "ref_fact", "ref_data"):
            getattr(L, nm).restype = dp
            getattr(L, nm).argtypes = [C.c_void_p]
        for nm in ("ref_fact_nd", "ref_data_nd", "ref_soln_nd", "ref_tree", "ref_cholfacts"):
            getattr(L, nm).restype = C.c_void_p
            getattr(L, nm).argtypes = [C.c_void_p]
        L.ref_nvars.argtypes = [C.c_void_p]
        L.ref_depth.argtypes = [C.c_void_p]
        L.ref_profile.argtypes = [C.c_void_p, dp]
        L.ref_factor_separator.argtypes = [C.c_void_p, C.c_int, C.c_int]
        L.ref_solve_chol_factor.argtypes = [C.c_void_p] + [C.c_int] * 3
        L.ref_bench.restype = C.c_double
        L.ref_bench.argtypes = [C.c_int] * 5 + [dp] * 8 + [C.c_int]
        # the reference's own stage functions, straight from its sources
        L.ndlqr_SolveLeaf.argtypes = [C.c_void_p, C.c_int]
        L.ndlqr_SolveLeaves.argtypes = [C.c_void_p]
        L.ndlqr_FactorInnerProduct.argtypes = [C.c_void_p, C.c_void_p] + [C.c_int] * 3
        L.ndlqr_UpdateShurFactor.argtypes = [C.c_void_p, C.c_void_p] + [C.c_int] * 4 + [C.c_bool]
        L.ndlqr_ComputeShurCompliment.argtypes = [C.c_void_p] + [C.c_int] * 3
        L.ndlqr_ShouldCalcLambda.restype = C.c_bool
        L.ndlqr_ShouldCalcLambda.argtypes = [C.c_void_p, C.c_int, C.c_int]
        L.ndlqr_GetIndexAtLevel.argtypes = [C.c_void_p, C.c_int, C.c_int]
        L.ndlqr_GetIndexLevel.argtypes = [C.c_void_p, C.c_int]
        L.ndlqr_GetIndexFromLeaf.argtypes = [C.c_void_p, C.c_int, C.c_int]

    def solver(self, prob):
        return RefSolver(self, prob)

    def riccati(self, prob, reps=1):
        """The reference's serial Riccati comparison solver (src/riccati_solve.c:7-150) on one problem:
        (solution [nvars] in the reference's [lambda x u] order, mean ms per solve). Fixtures only -- it
        diverges on random long horizons."""
        fn = self.L.ref_riccati_bench
        fn.restype = C.c_double
        fn.argtypes = [C.c_int] * 4 + [dp] * 8 + [dp, C.POINTER(C.c_int)]
        soln = np.zeros(prob.nvars)
        nv = C.c_int(0)
        ms = fn(prob.n, prob.m, prob.N, reps, *prob.cargs(), _p(soln), C.byref(nv))
        assert ms >= 0 and nv.value == prob.nvars, (ms, nv.value, prob.nvars)
        return soln, ms


class RefSolver:
    def __init__(self, ref, prob):
        self.L, self.prob = ref.L, prob
        self.h = self.L.ref_new_solver(prob.n, prob.m, prob.N, *prob.cargs())
        assert self.h
        n, m, N, K = prob.n, prob.m, prob.N, prob.K
        self.zb, self.fb = 2 * n + m, (2 * n + m) * n
        self._nf, self._ns = N * K * self.fb, N * self.zb

    def __del__(self):
        if getattr(self, "h", None):
            self.L.ref_free_solver(self.h)
            self.h = None

    def solve(self, nthreads=1):
        return self.L.ref_solve(self.h, nthreads)

    def fact(self):
        return np.ctypeslib.as_array(self.L.ref_fact(self.h), (self._nf,))

    def data(self):
        return np.ctypeslib.as_array(self.L.ref_data(self.h), (self._nf,))

    def soln(self):
        return np.ctypeslib.as_array(self.L.ref_soln(self.h), (self._ns,))


def kkt_residual_ld(prob, z):
    """b - K z of the reference's KKT system (src/solver.c:122-194) in extended precision (numpy longdouble: the
    80-bit x87 format on the hosts used here), for a solution z in the reference's [lambda x u] order. Returns
    (r_lam [N, n], r_x [N, n], r_u [N, m]) -- the residual rows of each knot. Test infrastructure."""
    ld = np.longdouble
    n, m, N = prob.n, prob.m, prob.N
    zb = 2 * n + m
    full = np.zeros(N * zb, dtype=ld)
    full[: z.size] = z
    Z = full.reshape(N, zb)
    lam, x, u = Z[:, :n], Z[:, n:2 * n], Z[:, 2 * n:]
    A = prob.A.astype(ld).reshape(N, n, n).transpose(0, 2, 1)  # column-major storage -> A[k][i, j]
    B = prob.B.astype(ld).reshape(N, m, n).transpose(0, 2, 1)
    Q, R = prob.Q.astype(ld), prob.R.astype(ld)
    q, r, d, x0 = prob.q.astype(ld), prob.r.astype(ld), prob.d.astype(ld), prob.x0.astype(ld)
    r_lam = np.zeros((N, n), dtype=ld)
    r_x = np.zeros((N, n), dtype=ld)
    r_u = np.zeros((N, m), dtype=ld)
    r_lam[0] = -x0 - (-x[0])
    for k in range(N):
        nxt = A[k].T @ lam[k + 1] if k < N - 1 else 0
        r_x[k] = -q[k] - (Q[k] * x[k] - lam[k] + nxt)
        if k < N - 1:
            r_u[k] = -r[k] - (R[k] * u[k] + B[k].T @ lam[k + 1])
            r_lam[k + 1] = -d[k] - (A[k] @ x[k] + B[k] @ u[k] - x[k + 1])
    return r_lam, r_x, r_u


def refined_solution(oracle, prob, iters=3):
    """The oracle's solution taken through `iters` steps of iterative refinement with the residual evaluated in
    extended precision: accurate to about double rounding even where the system is ill-conditioned. The yardstick
    for what "as accurate as the reference" means on hard inputs (the oracle's own error against it is the bar)."""
    z = oracle.solve(prob, 8)[0][: prob.nvars].astype(np.longdouble)
    for _ in range(iters):
        r_lam, r_x, r_u = kkt_residual_ld(prob, z)
        # K dz = res  <=>  a problem with the same A, B, Q, R and right-hand side (x0, q, r, d) = -(res rows)
        corr = Problem(prob.n, prob.m, prob.N, prob.A, prob.B, prob.Q, prob.R, (-r_x).astype(np.float64),
                       (-r_u).astype(np.float64), (-r_lam[1:]).astype(np.float64).tolist() + [[0.0] * prob.n],
                       (-r_lam[0]).astype(np.float64))
        z = z + oracle.solve(corr, 8)[0][: prob.nvars]
    return z.astype(np.float64)

"""ctypes mirror of include/ndlqr.h and include/ndlqr_hip.h.

Function names, argument order and return conventions are those of the C API (and therefore of
the reference's src/*.h); nothing numerical happens in Python. If the shared library is missing
it is built in-tree with hipcc (rslqr_amd.build); if that fails the import fails loudly -- there
is no Python or CPU fallback for the solver.
"""
import ctypes as C
import os
import re

import numpy as np

from . import build as _build

FLAG_STRICT_FP = 1
FLAG_GENERIC = 2
FLAG_PROFILE = 4
FLAG_KEEP_FACT = 8
FLAG_KEEP_RECORDS = 16
SOLN_LAMBDA, SOLN_STATE, SOLN_INPUT, SOLN_ONLY = 1, 2, 4, 8

ERR_INVALID = -1
ERR_NO_DEVICE = -2
ERR_NOT_SPD = -3

dp = C.POINTER(C.c_double)


class Matrix(C.Structure):
    _fields_ = [("rows", C.c_int), ("cols", C.c_int), ("data", dp)]

    def numpy(self):
        """Copy out as a (rows, cols) array (storage is column-major)."""
        flat = np.ctypeslib.as_array(self.data, (self.rows * self.cols,))
        return flat.reshape(self.cols, self.rows).T.copy()


class CholeskyInfo(C.Structure):
    _fields_ = [("uplo", C.c_char), ("success", C.c_int), ("lib", C.c_char), ("fact", C.c_void_p),
                ("is_freed", C.c_int)]


class LQRData(C.Structure):
    _fields_ = [("nstates", C.c_int), ("ninputs", C.c_int), ("Q", dp), ("R", dp), ("q", dp),
                ("r", dp), ("c", dp), ("A", dp), ("B", dp), ("d", dp)]


class LQRProblem(C.Structure):
    _fields_ = [("nhorizon", C.c_int), ("x0", dp), ("lqrdata", C.POINTER(C.POINTER(LQRData)))]


class UnitRange(C.Structure):
    _fields_ = [("start", C.c_int), ("stop", C.c_int)]


class BinaryNode(C.Structure):
    pass


BinaryNode._fields_ = [("idx", C.c_int), ("level", C.c_int), ("levelidx", C.c_int),
                       ("left_inds", UnitRange), ("right_inds", UnitRange),
                       ("parent", C.POINTER(BinaryNode)), ("left_child", C.POINTER(BinaryNode)),
                       ("right_child", C.POINTER(BinaryNode))]


class OrderedBinaryTree(C.Structure):
    _fields_ = [("root", C.POINTER(BinaryNode)), ("node_list", C.POINTER(BinaryNode)),
                ("num_elements", C.c_int), ("depth", C.c_int)]


class NdFactor(C.Structure):
    _fields_ = [("lambda_", Matrix), ("state", Matrix), ("input", Matrix)]


class NdData(C.Structure):
    _fields_ = [("nstates", C.c_int), ("ninputs", C.c_int), ("nsegments", C.c_int),
                ("depth", C.c_int), ("width", C.c_int), ("data", dp),
                ("factors", C.POINTER(NdFactor))]

    def numpy(self):
        """View of the whole slab (no copy)."""
        count = (self.nsegments + 1) * self.depth * (2 * self.nstates + self.ninputs) * self.width
        return np.ctypeslib.as_array(self.data, (count,))


class NdLqrCholeskyFactors(C.Structure):
    _fields_ = [("depth", C.c_int), ("nhorizon", C.c_int), ("cholinfo", C.POINTER(CholeskyInfo)),
                ("numfacts", C.c_int)]


class NdLqrProfile(C.Structure):
    _fields_ = [("t_total_ms", C.c_double), ("t_leaves_ms", C.c_double),
                ("t_products_ms", C.c_double), ("t_cholesky_ms", C.c_double),
                ("t_cholsolve_ms", C.c_double), ("t_shur_ms", C.c_double),
                ("num_threads", C.c_int)]


class NdLqrSolver(C.Structure):
    _fields_ = [("nstates", C.c_int), ("ninputs", C.c_int), ("nhorizon", C.c_int),
                ("depth", C.c_int), ("nvars", C.c_int), ("tree", OrderedBinaryTree),
                ("diagonals", C.POINTER(Matrix)), ("data", C.POINTER(NdData)),
                ("fact", C.POINTER(NdData)), ("soln", C.POINTER(NdData)),
                ("cholfacts", C.POINTER(NdLqrCholeskyFactors)), ("solve_time_ms", C.c_double),
                ("linalg_time_ms", C.c_double), ("profile", NdLqrProfile),
                ("num_threads", C.c_int), ("device_ctx", C.c_void_p),
                ("device_flags", C.c_uint), ("device_profiling", C.c_int), ("device_profiled", C.c_int),
                ("device_split", NdLqrProfile), ("mirror_fact", C.c_int)]


_LIB = None


def library_path():
    return _build.LIB


def exported_symbols():
    """Every function name declared in include/ndlqr.h and include/ndlqr_hip.h."""
    names = []
    for hdr in ("ndlqr.h", "ndlqr_hip.h"):
        text = open(os.path.join(_build.INCLUDE, hdr)).read()
        text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
        text = re.sub(r"static inline[^{]*\{[^}]*\}", "", text)
        for m in re.finditer(r"\b([A-Za-z_][A-Za-z0-9_]*)\s*\([^;{]*\)\s*;", text):
            names.append(m.group(1))
    return sorted(set(names))


def lib():
    """Load (building if needed) librslqr_amd.so and declare the prototypes used from Python."""
    global _LIB
    if _LIB is not None:
        return _LIB
    if not os.path.exists(_build.LIB):
        _build.build()
    # NDLQR_LIBRARY: developer hook for instrumented builds of the same library (tools/segtime.py)
    L = C.CDLL(os.environ.get("NDLQR_LIBRARY", _build.LIB))
    vp, ci, cd, cu64 = C.c_void_p, C.c_int, C.c_double, C.c_uint64
    sp = C.POINTER(NdLqrSolver)
    pp = C.POINTER(LQRProblem)
    ndp = C.POINTER(NdData)
    mp = C.POINTER(Matrix)

    def proto(name, restype, *argtypes):
        fn = getattr(L, name)
        fn.restype = restype
        fn.argtypes = list(argtypes)

    proto("ndlqr_Version", C.c_char_p)
    proto("ndlqr_hip_device_count", ci)
    proto("ndlqr_hip_last_error", C.c_char_p)
    # problem containers
    proto("ndlqr_NewLQRData", C.POINTER(LQRData), ci, ci)
    proto("ndlqr_FreeLQRData", ci, C.POINTER(LQRData))
    proto("ndlqr_InitializeLQRData", ci, C.POINTER(LQRData), dp, dp, dp, dp, cd, dp, dp, dp)
    proto("ndlqr_CopyLQRData", ci, C.POINTER(LQRData), C.POINTER(LQRData))
    proto("ndlqr_NewLQRProblem", pp, ci, ci, ci)
    proto("ndlqr_InitializeLQRProblem", ci, pp, dp, C.POINTER(C.POINTER(LQRData)))
    proto("ndlqr_FreeLQRProblem", ci, pp)
    proto("ndlqr_ReadLQRProblemJSONFile", pp, C.c_char_p)
    proto("ndlqr_ReadLQRDataJSONFile", C.POINTER(LQRData), C.c_char_p)
    proto("ReadMatrixJSONFile", Matrix, C.c_char_p, C.c_char_p)
    proto("FreeMatrix", ci, mp)
    proto("ReadFile", ci, C.c_char_p, C.POINTER(C.c_char_p), C.POINTER(ci))
    proto("ndlqr_NewSyntheticLQRProblem", pp, ci, ci, ci, cu64)
    proto("ndlqr_GenerateSyntheticFlat", ci, ci, ci, ci, cu64, dp, dp, dp, dp, dp, dp, dp, dp)
    # tree / storage
    proto("ndlqr_BuildTree", OrderedBinaryTree, ci)
    proto("ndlqr_FreeTree", ci, C.POINTER(OrderedBinaryTree))
    proto("ndlqr_GetIndexFromLeaf", ci, C.POINTER(OrderedBinaryTree), ci, ci)
    proto("ndlqr_GetIndexLevel", ci, C.POINTER(OrderedBinaryTree), ci)
    proto("ndlqr_GetIndexAtLevel", ci, C.POINTER(OrderedBinaryTree), ci, ci)
    proto("ndlqr_NewNdData", ndp, ci, ci, ci, ci)
    proto("ndlqr_FreeNdData", ci, ndp)
    proto("ndlqr_ResetNdData", None, ndp)
    proto("ndlqr_GetNdFactor", ci, ndp, ci, ci, C.POINTER(C.POINTER(NdFactor)))
    proto("ndlqr_NewCholeskyFactors", C.POINTER(NdLqrCholeskyFactors), ci, ci)
    proto("ndlqr_FreeCholeskyFactors", ci, C.POINTER(NdLqrCholeskyFactors))
    proto("ndlqr_GetSFactorization", ci, C.POINTER(NdLqrCholeskyFactors), ci, ci,
          C.POINTER(C.POINTER(CholeskyInfo)))
    # solver
    proto("ndlqr_NewNdLqrSolver", sp, ci, ci, ci)
    proto("ndlqr_FreeNdLqrSolver", ci, sp)
    proto("ndlqr_InitializeWithLQRProblem", ci, pp, sp)
    proto("ndlqr_ResetSolver", None, sp)
    proto("ndlqr_GetNumVars", ci, sp)
    proto("ndlqr_SetNumThreads", ci, sp, ci)
    proto("ndlqr_GetNumThreads", ci, sp)
    proto("ndlqr_PrintSolveSummary", None, sp)
    proto("ndlqr_PrintSolveProfile", ci, sp)
    proto("ndlqr_GetProfile", NdLqrProfile, sp)
    proto("ndlqr_Solve", ci, sp)
    proto("ndlqr_GetSolution", Matrix, sp)
    proto("ndlqr_CopySolution", ci, sp, dp)
    proto("ndlqr_SyncFactorsToHost", ci, sp)
    proto("ndlqr_SetDeviceProfiling", ci, sp, ci)
    proto("ndlqr_SetDeviceFlags", ci, sp, C.c_uint)
    proto("ndlqr_SetFactorMirroring", ci, sp, ci)
    # stage functions
    proto("ndlqr_SolveLeaf", ci, sp, ci)
    proto("ndlqr_SolveLeaves", ci, sp)
    proto("ndlqr_FactorInnerProduct", ci, ndp, ndp, ci, ci, ci)
    proto("ndlqr_SolveCholeskyFactor", ci, ndp, C.POINTER(CholeskyInfo), ci, ci, ci)
    proto("ndlqr_UpdateShurFactor", ci, ndp, ndp, ci, ci, ci, ci, C.c_bool)
    proto("ndlqr_ShouldCalcLambda", C.c_bool, C.POINTER(OrderedBinaryTree), ci, ci)
    proto("ndlqr_ComputeShurCompliment", ci, sp, ci, ci, ci)
    # dense helpers
    proto("MatrixMultiply", None, mp, mp, mp, C.c_bool, C.c_bool, cd, cd)
    proto("MatrixCholeskyFactorize", ci, mp)
    proto("MatrixCholeskySolve", ci, mp, mp)
    proto("MatrixSymmetricMultiply", None, mp, mp, mp, cd, cd)
    proto("MatrixAddition", ci, mp, mp, cd)
    # batch
    proto("ndlqr_NewBatchSolver", vp, ci, ci, ci, ci, ci)
    proto("ndlqr_FreeBatchSolver", ci, vp)
    proto("ndlqr_BatchSetFlags", ci, vp, C.c_uint)
    proto("ndlqr_BatchGetFlags", C.c_uint, vp)
    proto("ndlqr_InitializeBatch", ci, vp, C.POINTER(pp), ci)
    proto("ndlqr_InitializeBatchFlat", ci, vp, dp, dp, dp, dp, dp, dp, dp, dp)
    proto("ndlqr_InitializeBatchSynthetic", ci, vp, cu64)
    proto("ndlqr_InitializeBatchFlatDevice", ci, vp, vp, vp, vp, vp, vp, vp, vp, vp)
    proto("ndlqr_SolveBatch", ci, vp)
    proto("ndlqr_BatchSetRhsFlat", ci, vp, dp, dp, dp, dp)
    proto("ndlqr_SolveBatchRhsOnly", ci, vp)
    proto("ndlqr_SolveBatchMultiRhs", ci, vp, ci, dp, dp, dp, dp, dp)
    proto("ndlqr_SolveBatchMultiRhsSlices", ci, vp, ci, dp, dp, dp, dp, ci, ci, C.c_uint, dp)
    proto("ndlqr_SolveBatchAsync", ci, vp)
    proto("ndlqr_BatchStepAsync", ci, vp, dp, dp, dp, dp, dp)
    proto("ndlqr_BatchSynchronizePrevious", ci, vp)
    proto("ndlqr_BatchSetStepSelection", ci, vp, ci, ci, C.c_uint)
    proto("ndlqr_CopyBatchSolutionSlices", ci, vp, ci, ci, C.c_uint, dp)
    proto("ndlqr_SolveBatchSlicesAsync", ci, vp, ci, ci, C.c_uint, dp)
    proto("ndlqr_BatchTimeShardTopDoubles", ci, vp, ci)
    proto("ndlqr_BatchTimeShardFactor", ci, vp, ci, ci)
    proto("ndlqr_BatchTimeShardExportTop", ci, vp, ci, vp)
    proto("ndlqr_BatchTimeShardImportTop", ci, vp, ci, vp)
    proto("ndlqr_BatchTimeShardFinish", ci, vp, ci, ci)
    proto("ndlqr_HostAlloc", vp, C.c_size_t)
    proto("ndlqr_HostFree", None, vp)
    proto("ndlqr_DeviceAlloc", vp, C.c_size_t)
    proto("ndlqr_DeviceFree", None, vp)
    proto("ndlqr_DeviceCopy", ci, vp, vp, C.c_size_t)
    proto("ndlqr_BatchSynchronize", ci, vp)
    proto("ndlqr_BatchNumVars", ci, vp)
    proto("ndlqr_BatchSize", ci, vp)
    proto("ndlqr_CopyBatchSolution", ci, vp, ci, dp)
    proto("ndlqr_CopyBatchSolutions", ci, vp, dp)
    proto("ndlqr_CopyBatchSolutionsDevice", ci, vp, vp)
    proto("ndlqr_CopyBatchFactors", ci, vp, ci, dp)
    proto("ndlqr_BatchCholeskyFailures", ci, vp)
    proto("ndlqr_BatchKktResiduals", ci, vp, dp, dp)
    proto("ndlqr_BatchSolveTimeMs", cd, vp)
    proto("ndlqr_BatchDeviceContext", vp, vp)
    # shim bits used by the benchmark
    proto("ndlqr_hip_set_stream", ci, vp, vp)
    proto("ndlqr_hip_get_stream", vp, vp)
    proto("ndlqr_hip_profile_slots", ci, vp)
    proto("ndlqr_hip_profile_get", ci, vp, ci, C.c_char_p, ci, dp, C.POINTER(ci))
    proto("ndlqr_hip_profile_reset", ci, vp)
    proto("ndlqr_hip_device_pointers", ci, vp, C.POINTER(vp))
    proto("ndlqr_hip_upload_inputs", ci, vp, ci, ci, dp, dp, dp)
    proto("ndlqr_hip_factors_valid", ci, vp)
    proto("ndlqr_hip_schedule", C.c_char_p, vp)
    proto("ndlqr_hip_set_pipeline_depth", ci, vp, ci)
    proto("ndlqr_hip_pipeline_depth", ci, vp)
    proto("ndlqr_hip_pack_solutions_device", ci, vp, vp)
    proto("ndlqr_hip_gemm", ci, ci, ci, ci, ci, ci, cd, dp, ci, dp, ci, cd, dp, ci)
    proto("ndlqr_hip_potrf_lower", ci, ci, dp, ci)
    proto("ndlqr_hip_potrs_lower", ci, ci, ci, dp, ci, dp, ci)
    _LIB = L
    return L


def pinned_empty(shape):
    """float64 numpy array in pinned host memory (ndlqr_HostAlloc): H2D / D2H copies from / to it are
    asynchronous and run at the rate of the host link. Freed when the array is garbage-collected."""
    import weakref
    count = int(np.prod(shape))
    L = lib()
    ptr = L.ndlqr_HostAlloc(max(count, 1) * 8)
    if not ptr:
        raise MemoryError("ndlqr_HostAlloc(%d bytes) failed" % (count * 8))
    buf = (C.c_double * max(count, 1)).from_address(ptr)
    arr = np.ctypeslib.as_array(buf)[:count].reshape(shape)
    weakref.finalize(buf, L.ndlqr_HostFree, C.c_void_p(ptr))
    return arr


class DeviceArray:
    """float64 values of `shape` in device memory (ndlqr_DeviceAlloc): what BatchSolver.step_async takes for q, r, d, x0
    and soln when the loop around the solver lives on the GPU -- nothing crosses the host link then. set() / get() copy
    from / to numpy arrays synchronously (ndlqr_DeviceCopy); `ptr` is the raw address for other device code."""

    def __init__(self, shape):
        import weakref
        self.shape = tuple(int(s) for s in (shape if isinstance(shape, (tuple, list)) else (shape,)))
        self.size = int(np.prod(self.shape))
        L = lib()
        self.ptr = L.ndlqr_DeviceAlloc(max(self.size, 1) * 8)
        if not self.ptr:
            raise MemoryError("ndlqr_DeviceAlloc(%d bytes) failed" % (self.size * 8))
        weakref.finalize(self, L.ndlqr_DeviceFree, C.c_void_p(self.ptr))

    def set(self, arr):
        arr = np.ascontiguousarray(arr, dtype=np.float64)
        assert arr.size == self.size
        if lib().ndlqr_DeviceCopy(C.c_void_p(self.ptr), arr.ctypes.data_as(C.c_void_p), self.size * 8):
            raise RuntimeError("ndlqr_DeviceCopy failed")
        return self

    def get(self):
        out = np.empty(self.shape)
        if lib().ndlqr_DeviceCopy(out.ctypes.data_as(C.c_void_p), C.c_void_p(self.ptr), self.size * 8):
            raise RuntimeError("ndlqr_DeviceCopy failed")
        return out


def device_count():
    return lib().ndlqr_hip_device_count()


def _ptr(a):
    return a.ctypes.data_as(dp)


def generate_synthetic(n, m, N, seed):
    """ndlqr_GenerateSyntheticFlat -> dict of numpy arrays (A [N,n*n] col-major, ...)."""
    out = dict(A=np.zeros((N, n * n)), B=np.zeros((N, n * m)), Q=np.zeros((N, n)),
               R=np.zeros((N, m)), q=np.zeros((N, n)), r=np.zeros((N, m)), d=np.zeros((N, n)),
               x0=np.zeros(n))
    err = lib().ndlqr_GenerateSyntheticFlat(n, m, N, seed, *[_ptr(out[k]) for k in
                                                               ("A", "B", "Q", "R", "q", "r", "d", "x0")])
    if err:
        raise RuntimeError("ndlqr_GenerateSyntheticFlat failed: %d" % err)
    return out


class BatchSolver:
    """numpy-friendly wrapper of NdLqrBatchSolver (include/ndlqr.h, batch API)."""

    def __init__(self, n, m, N, batch, device=-1, flags=0):
        self.L = lib()
        self.n, self.m, self.N, self.batch = n, m, N, batch
        self.h = self.L.ndlqr_NewBatchSolver(n, m, N, batch, device)
        if not self.h:
            raise RuntimeError("ndlqr_NewBatchSolver failed: %s" %
                               self.L.ndlqr_hip_last_error().decode())
        self.nvars = self.L.ndlqr_BatchNumVars(self.h)
        self._sel = None          # (knot0, nknots, blocks) of set_step_selection
        self._step_refs = []      # host arrays of the steps in flight (kept alive until they are synchronised)
        if flags:
            self.set_flags(flags)

    def close(self):
        if getattr(self, "h", None):
            self.L.ndlqr_FreeBatchSolver(self.h)  # (waits for everything in flight)
            self.h = None
        self._step_refs = []

    __del__ = close

    def set_flags(self, flags):
        self.L.ndlqr_BatchSetFlags(self.h, flags)

    @property
    def ctx(self):
        return self.L.ndlqr_BatchDeviceContext(self.h)

    def initialize_flat(self, A, B, Q, R, q, r, d, x0):
        n, m, N, bt = self.n, self.m, self.N, self.batch
        shapes = dict(A=(bt, N, n * n), B=(bt, N, n * m), Q=(bt, N, n), R=(bt, N, m),
                      q=(bt, N, n), r=(bt, N, m), d=(bt, N, n), x0=(bt, n))
        arrs = []
        for name, a in zip(("A", "B", "Q", "R", "q", "r", "d", "x0"), (A, B, Q, R, q, r, d, x0)):
            a = np.ascontiguousarray(a, dtype=np.float64)
            if a.size != int(np.prod(shapes[name])):
                raise ValueError("bad size for %s" % name)
            arrs.append(a)
        err = self.L.ndlqr_InitializeBatchFlat(self.h, *[_ptr(a) for a in arrs])
        if err:
            raise RuntimeError("ndlqr_InitializeBatchFlat failed: %d" % err)

    def initialize_flat_device(self, *device_ptrs):
        """Eight device pointers (ints), flat reference layout: A, B, Q, R, q, r, d, x0."""
        err = self.L.ndlqr_InitializeBatchFlatDevice(self.h, *[C.c_void_p(int(p)) for p in device_ptrs])
        if err:
            raise RuntimeError("ndlqr_InitializeBatchFlatDevice failed: %d" % err)

    def initialize_synthetic(self, seed0):
        err = self.L.ndlqr_InitializeBatchSynthetic(self.h, seed0)
        if err:
            raise RuntimeError("ndlqr_InitializeBatchSynthetic failed: %d" % err)

    def solve(self):
        return self.L.ndlqr_SolveBatch(self.h)

    def set_rhs_flat(self, q, r, d, x0):
        arrs = [np.ascontiguousarray(a, dtype=np.float64) for a in (q, r, d, x0)]
        err = self.L.ndlqr_BatchSetRhsFlat(self.h, *[_ptr(a) for a in arrs])
        if err:
            raise RuntimeError("ndlqr_BatchSetRhsFlat failed: %d" % err)

    def solve_rhs_only(self):
        """Solution sweep against the cached factorisation (needs FLAG_KEEP_FACT, or FLAG_KEEP_RECORDS
        in fast mode on a size-specialised shape, on the solve)."""
        return self.L.ndlqr_SolveBatchRhsOnly(self.h)

    def solve_multi_rhs(self, q, r, d, x0, out=None, selection=None):
        """nrhs sets of right-hand sides for the whole batch against the records kept by the last solve (FLAG_KEEP_RECORDS,
        level-per-launch schedule): q, d [nrhs][batch][N][n], r [nrhs][batch][N][m], x0 [nrhs][batch][n] ->
        solutions [nrhs][batch][nvars], or with selection = (knot0, nknots, blocks) that slice of every solution,
        [nrhs][batch][nknots][width] (ndlqr_SolveBatchMultiRhsSlices: nothing else is computed by the last launch).
        Blocking; solve_ms() afterwards = device time of the solve kernels alone."""
        arrs = [np.ascontiguousarray(a, dtype=np.float64) for a in (q, r, d, x0)]
        nrhs = arrs[3].shape[0]
        assert arrs[0].shape == (nrhs, self.batch, self.N, self.n) and arrs[1].shape == (nrhs, self.batch, self.N, self.m)
        assert arrs[2].shape == (nrhs, self.batch, self.N, self.n) and arrs[3].shape == (nrhs, self.batch, self.n)
        if selection is not None:
            k0, nk, blocks = selection
            if out is None:
                out = np.empty((nrhs, self.batch, nk, self.slice_width(blocks)))
            assert out.size == nrhs * self.batch * nk * self.slice_width(blocks)
            err = self.L.ndlqr_SolveBatchMultiRhsSlices(self.h, nrhs, *[_ptr(a) for a in arrs], k0, nk, blocks, _ptr(out))
        else:
            if out is None:
                out = np.empty((nrhs, self.batch, self.nvars))
            err = self.L.ndlqr_SolveBatchMultiRhs(self.h, nrhs, *[_ptr(a) for a in arrs], _ptr(out))
        if err:
            raise RuntimeError("ndlqr_SolveBatchMultiRhs failed: %d (%s)" % (err, self.L.ndlqr_hip_last_error().decode()))
        return out

    def solve_async(self):
        return self.L.ndlqr_SolveBatchAsync(self.h)

    def slice_width(self, blocks):
        return (self.n if blocks & SOLN_LAMBDA else 0) + (self.n if blocks & SOLN_STATE else 0) + \
               (self.m if blocks & SOLN_INPUT else 0)

    def set_step_selection(self, knot0=0, nknots=0, blocks=7):
        """ndlqr_BatchSetStepSelection: what step_async brings down -- knots [knot0, knot0 + nknots), blocks = SOLN_*
        mask, packed [batch, nknots, width]; nknots = 0: every solution [batch, nvars] (default). With SOLN_ONLY in the
        mask a step computes nothing but those knots (the rest of the solution is unavailable until the next complete
        solve)."""
        err = self.L.ndlqr_BatchSetStepSelection(self.h, knot0, nknots, blocks)
        if err:
            raise ValueError("ndlqr_BatchSetStepSelection(%d, %d, %d): %d" % (knot0, nknots, blocks, err))
        self._sel = (knot0, nknots, blocks) if nknots else None

    def solve_slices_async(self, knot0, nknots, blocks, out):
        """ndlqr_SolveBatchSlicesAsync: factor + solve of the resident problems, computing and delivering knots
        [knot0, knot0 + nknots) alone into `out` ([batch, nknots, width]: a pinned_empty array or a DeviceArray);
        complete after synchronize()."""
        size = self.batch * nknots * self.slice_width(blocks)
        assert out.size == size and (isinstance(out, DeviceArray) or (out.dtype == np.float64 and out.flags["C_CONTIGUOUS"]))
        self._step_refs = self._step_refs[-1:] + [(out,)]
        ptr = C.cast(C.c_void_p(out.ptr), dp) if isinstance(out, DeviceArray) else _ptr(out)
        return self.L.ndlqr_SolveBatchSlicesAsync(self.h, knot0, nknots, blocks, ptr)

    def solution_slices(self, knot0, nknots, blocks, out=None):
        """ndlqr_CopyBatchSolutionSlices: [batch, nknots, width] of the latest solve."""
        if out is None:
            out = np.zeros((self.batch, nknots, self.slice_width(blocks)))
        assert out.dtype == np.float64 and out.flags["C_CONTIGUOUS"] and out.size == self.batch * nknots * self.slice_width(blocks)
        err = self.L.ndlqr_CopyBatchSolutionSlices(self.h, knot0, nknots, blocks, _ptr(out))
        if err:
            raise RuntimeError("ndlqr_CopyBatchSolutionSlices failed: %d" % err)
        return out

    def step_async(self, q, r, d, x0, soln):
        """ndlqr_BatchStepAsync: new right-hand side up, factor + solve, solutions down into `soln` ([batch, nvars], or
        the slice chosen with set_step_selection), asynchronously. Use pinned_empty() arrays (pageable ones make the
        call block) or DeviceArray objects (no transfer at all) and leave them untouched until the step has been synchronised; the solver holds references to the
        arrays of the two steps that can be in flight, so that dropping one early does not free pinned memory the GPU
        is still reading or writing."""
        n, m, N, bt = self.n, self.m, self.N, self.batch
        out_size = bt * self.nvars if self._sel is None else bt * self._sel[1] * self.slice_width(self._sel[2])
        for a, size in ((q, bt * N * n), (r, bt * N * m), (d, bt * N * n), (x0, bt * n), (soln, out_size)):
            assert a is None or (a.size == size if isinstance(a, DeviceArray) else
                                 (a.dtype == np.float64 and a.flags["C_CONTIGUOUS"] and a.size == size))
        assert x0 is not None and soln is not None  # q, r, d may be None: unchanged
        ptr = lambda a: None if a is None else (C.cast(C.c_void_p(a.ptr), dp) if isinstance(a, DeviceArray) else _ptr(a))
        self._step_refs = self._step_refs[-1:] + [(q, r, d, x0, soln)]
        return self.L.ndlqr_BatchStepAsync(self.h, ptr(q), ptr(r), ptr(d), ptr(x0), ptr(soln))

    # ---- time-axis sharding (one problem over G ranks; include/ndlqr_hip.h)
    def time_shard_top_doubles(self, G):
        return self.L.ndlqr_BatchTimeShardTopDoubles(self.h, G)

    def time_shard_factor(self, g, G):
        return self.L.ndlqr_BatchTimeShardFactor(self.h, g, G)

    def time_shard_export(self, G, ptr):
        """`ptr`: address (int) of host or device memory for time_shard_top_doubles(G) doubles."""
        return self.L.ndlqr_BatchTimeShardExportTop(self.h, G, C.c_void_p(int(ptr)))

    def time_shard_import(self, G, ptr):
        return self.L.ndlqr_BatchTimeShardImportTop(self.h, G, C.c_void_p(int(ptr)))

    def time_shard_finish(self, g, G):
        return self.L.ndlqr_BatchTimeShardFinish(self.h, g, G)

    def synchronize_previous(self):
        return self.L.ndlqr_BatchSynchronizePrevious(self.h)

    def synchronize(self):
        err = self.L.ndlqr_BatchSynchronize(self.h)
        self._step_refs = []
        return err

    def solve_ms(self):
        return self.L.ndlqr_BatchSolveTimeMs(self.h)

    def solution(self, p):
        out = np.zeros(self.nvars)
        got = self.L.ndlqr_CopyBatchSolution(self.h, p, _ptr(out))
        if got != self.nvars:
            raise RuntimeError("ndlqr_CopyBatchSolution failed: %d" % got)
        return out

    def kkt_residuals(self):
        """(res, bnorm): ||K z - b||_2 and ||b||_2 of every problem, evaluated on the device."""
        res = np.zeros(self.batch)
        bn = np.zeros(self.batch)
        err = self.L.ndlqr_BatchKktResiduals(self.h, _ptr(res), _ptr(bn))
        if err:
            raise RuntimeError("ndlqr_BatchKktResiduals failed: %d" % err)
        return res, bn

    def solutions(self, out=None):
        """[batch, nvars] solutions of the latest solve; `out`: destination (e.g. a pinned_empty array)."""
        if out is None:
            out = np.zeros((self.batch, self.nvars))
        assert out.dtype == np.float64 and out.flags["C_CONTIGUOUS"] and out.size == self.batch * self.nvars
        got = self.L.ndlqr_CopyBatchSolutions(self.h, _ptr(out))
        if got != self.nvars:
            raise RuntimeError("ndlqr_CopyBatchSolutions failed: %d" % got)
        return out

    def solutions_to_device(self, device_ptr):
        """[batch][nvars] packed solutions into device memory (asynchronous on the solver's stream)."""
        got = self.L.ndlqr_CopyBatchSolutionsDevice(self.h, C.c_void_p(int(device_ptr)))
        if got != self.nvars:
            raise RuntimeError("ndlqr_CopyBatchSolutionsDevice failed: %d" % got)

    def upload_packed(self, AB, QR, rhs):
        """Raw H2D of inputs already in the device layout of include/ndlqr_hip.h (whole batch)."""
        err = self.L.ndlqr_hip_upload_inputs(self.ctx, 0, self.batch, _ptr(AB), _ptr(QR), _ptr(rhs))
        if err:
            raise RuntimeError("ndlqr_hip_upload_inputs failed: %d" % err)

    def factors(self, p):
        K = int(np.log2(self.N))
        out = np.zeros(self.N * K * (2 * self.n + self.m) * self.n)
        err = self.L.ndlqr_CopyBatchFactors(self.h, p, _ptr(out))
        if err:
            raise RuntimeError("ndlqr_CopyBatchFactors failed: %d" % err)
        return out

    def cholesky_failures(self):
        return self.L.ndlqr_BatchCholeskyFailures(self.h)

    def profile(self):
        """{kernel name: (total ms, launches)} accumulated since the last reset."""
        out = {}
        for slot in range(self.L.ndlqr_hip_profile_slots(self.ctx)):
            name = C.create_string_buffer(64)
            ms, cnt = C.c_double(0), C.c_int(0)
            self.L.ndlqr_hip_profile_get(self.ctx, slot, name, 64, C.byref(ms), C.byref(cnt))
            out[name.value.decode()] = (ms.value, cnt.value)
        return out

    def schedule(self):
        """Name of the launch sequence the last solve used."""
        return self.L.ndlqr_hip_schedule(self.ctx).decode()

    def set_pipeline_depth(self, depth):
        """1: stream-ordered solves; 2: consecutive asynchronous solves alternate between two buffer sets."""
        return self.L.ndlqr_hip_set_pipeline_depth(self.ctx, depth)

    def pipeline_depth(self):
        return self.L.ndlqr_hip_pipeline_depth(self.ctx)

    def profile_reset(self):
        self.L.ndlqr_hip_profile_reset(self.ctx)

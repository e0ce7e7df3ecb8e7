"""Developer tool (GPU box): the headline step of two builds of the library on ONE box through the entry points both have
(ndlqr_NewBatchSolver, ndlqr_InitializeBatchSynthetic, ndlqr_SolveBatchAsync, ndlqr_BatchSynchronize) -- for comparing
across rounds, where bench.py's ctypes mirror no longer loads the older library.
    python tools/ab_across_rounds.py rslqr_amd/librslqr_amd_prev.so rslqr_amd/librslqr_amd.so [n m N batch]"""
import ctypes as C
import subprocess
import sys
import time

if len(sys.argv) > 1 and sys.argv[1] == "--one":
    lib, n, m, N, batch = sys.argv[2], *map(int, sys.argv[3:7])
    L = C.CDLL(lib)
    L.ndlqr_NewBatchSolver.restype = C.c_void_p
    L.ndlqr_NewBatchSolver.argtypes = [C.c_int] * 5
    L.ndlqr_InitializeBatchSynthetic.argtypes = [C.c_void_p, C.c_uint64]
    L.ndlqr_SolveBatchAsync.argtypes = [C.c_void_p]
    L.ndlqr_BatchSynchronize.argtypes = [C.c_void_p]
    bs = L.ndlqr_NewBatchSolver(n, m, N, batch, 0)
    assert bs and L.ndlqr_InitializeBatchSynthetic(bs, 1) == 0
    for _ in range(150):  # (captures, second buffer set, clocks)
        L.ndlqr_SolveBatchAsync(bs)
    L.ndlqr_BatchSynchronize(bs)
    best = 1e9
    for _ in range(5):
        t0 = time.perf_counter()
        for _ in range(100):
            assert L.ndlqr_SolveBatchAsync(bs) == 0
        assert L.ndlqr_BatchSynchronize(bs) == 0
        best = min(best, (time.perf_counter() - t0) / 100)
    print("%.4f" % (best * 1e3))
    sys.exit(0)

libs = sys.argv[1:3]
shape = sys.argv[3:7] or ["12", "4", "256", "1024"]
for rep in range(3):
    row = []
    for lib in libs:
        out = subprocess.run([sys.executable, __file__, "--one", lib] + shape, capture_output=True, text=True)
        row.append(out.stdout.strip() or ("failed: " + out.stderr[-200:]))
    print("(%s) x %s  ms per step, best of five 100-step regions:  %s" % (",".join(shape[:3]), shape[3], "   ".join(
        "%s %s" % (lib.split("/")[-1], v) for lib, v in zip(libs, row))), flush=True)

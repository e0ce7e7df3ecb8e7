"""Developer tool (GPU box): ndlqr_SolveBatchMultiRhs -- device time of the solve kernels for batch x nrhs right-hand sides
against batch kept factorisations, (12,4,256) and (6,3,256)."""
import os
import sys
import time

import numpy as np

os.environ["NDLQR_TREE"] = "0"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import rslqr_amd as R  # noqa: E402

rng = np.random.default_rng(1)
for (n, m, N) in [(12, 4, 256), (6, 3, 256)]:
    for (batch, nrhs) in [(1, 1024), (4, 256), (16, 64), (64, 16), (1024, 1), (1, 8192)]:
        bs = R.BatchSolver(n, m, N, batch, flags=R.FLAG_KEEP_RECORDS)
        bs.initialize_synthetic(1)
        bs.solve()
        q, d = rng.standard_normal((nrhs, batch, N, n)), 0.1 * rng.standard_normal((nrhs, batch, N, n))
        r, x0 = rng.standard_normal((nrhs, batch, N, m)), rng.standard_normal((nrhs, batch, n))
        out = np.empty((nrhs, batch, bs.nvars))
        bs.solve_multi_rhs(q, r, d, x0, out=out)
        best, wall = 1e9, 1e9
        for _ in range(3):
            t0 = time.perf_counter()
            bs.solve_multi_rhs(q, r, d, x0, out=out)
            wall = min(wall, (time.perf_counter() - t0) * 1e3)
            best = min(best, bs.solve_ms())
        print((n, m, N), "problems %5d x rhs %5d: kernels %.3f ms = %.2f M solves/s (call incl. pageable transfers %.1f ms)" % (
            batch, nrhs, best, batch * nrhs / best / 1e3, wall), flush=True)
        bs.close()

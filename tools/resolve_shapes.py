"""Developer tool (GPU box): full solve (default flags, two in flight) against the right-hand-side re-solve on the compact
records (NDLQR_FLAG_KEEP_RECORDS), per shape, ms per 1024-problem batch."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import rslqr_amd as R  # noqa: E402


def timed(fn, sync, reps=40):
    for _ in range(6):
        fn()
    sync()
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        sync()
        best = min(best, (time.perf_counter() - t0) / reps * 1e3)
    return best


for (n, m, N, batch) in [(12, 4, 256, 1024), (6, 3, 256, 1024), (8, 4, 256, 1024), (9, 3, 256, 1024), (10, 4, 256, 1024),
                         (13, 4, 256, 1024), (12, 4, 1024, 512), (11, 3, 256, 1024)]:
    bs = R.BatchSolver(n, m, N, batch)
    bs.initialize_synthetic(1)
    full = timed(bs.solve_async, bs.synchronize)
    bs.close()
    bs = R.BatchSolver(n, m, N, batch, flags=R.FLAG_KEEP_RECORDS)
    bs.initialize_synthetic(1)
    keep = timed(bs.solve_async, bs.synchronize, reps=10)
    sched = bs.schedule()
    bs.solve()
    re = timed(bs.solve_rhs_only, lambda: None, reps=20)
    res, bn = bs.kkt_residuals()
    print((n, m, N, batch), "full %.3f ms | KEEP_RECORDS solve %.3f (%s) | re-solve %.3f ms (%.2f M/s) kkt %.1e" % (
        full, keep, sched, re, batch / re / 1e3, (res / np.maximum(1, bn)).max()), flush=True)
    bs.close()

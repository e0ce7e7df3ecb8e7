"""Developer tool (GPU box): block sizes beyond 128 states -- schedule, time, KKT residual; fast / strict / KEEP_FACT."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import rslqr_amd as R  # noqa: E402

for (n, m, N, batch) in [(128, 16, 64, 8), (144, 16, 64, 8), (150, 10, 64, 8), (160, 16, 64, 8), (192, 16, 64, 4), (200, 8, 32, 4),
                         (256, 32, 32, 4)]:
    for flags in (0, R.FLAG_STRICT_FP, R.FLAG_KEEP_FACT):
        try:
            bs = R.BatchSolver(n, m, N, batch, flags=flags)
            bs.initialize_synthetic(3)
            rc = bs.solve()
            for _ in range(2):  # (both buffer sets of the pipeline and their captured sequences exist before the clock starts)
                bs.solve_async()
            bs.synchronize()
            ms = 1e9
            for _ in range(3):
                t0 = time.perf_counter()
                for _ in range(2):
                    bs.solve_async()
                bs.synchronize()
                ms = min(ms, (time.perf_counter() - t0) / 2 * 1e3)
            res, bn = bs.kkt_residuals()
            print((n, m, N, batch), "flags", flags, bs.schedule(), "rc", rc, "%.2f ms per step, %.1f solves/s" % (ms, batch / ms * 1e3),
                  "kkt %.1e" % (res / np.maximum(1, bn)).max(), flush=True)
            bs.close()
        except Exception as e:  # noqa: BLE001
            print((n, m, N, batch), "flags", flags, "failed:", e, flush=True)

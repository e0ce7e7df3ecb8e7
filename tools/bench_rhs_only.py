"""Developer tool (GPU box): factor once with KEEP_FACT, then time new right-hand sides."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import rslqr_amd

n, m, N, batch = 12, 4, 256, 1024
for name, keep in (("records only", rslqr_amd.FLAG_KEEP_RECORDS), ("records + factor array", rslqr_amd.FLAG_KEEP_FACT),
                   ("strict level sweep", rslqr_amd.FLAG_KEEP_FACT | rslqr_amd.FLAG_STRICT_FP)):
    bs = rslqr_amd.BatchSolver(n, m, N, batch, flags=keep)
    bs.initialize_synthetic(1)
    bs.solve()
    full = []
    for _ in range(5):
        t0 = time.perf_counter(); bs.solve(); full.append(time.perf_counter() - t0)
    rhs = []
    for _ in range(3):
        bs.solve_rhs_only()
    for _ in range(10):
        t0 = time.perf_counter(); bs.solve_rhs_only(); rhs.append(time.perf_counter() - t0)
    bs.set_flags(keep | rslqr_amd.FLAG_PROFILE)
    bs.solve()
    bs.profile_reset(); bs.solve_rhs_only()
    print("%-12s factor+solve %.3f ms   rhs-only %.3f ms   %s" % (
        name, min(full) * 1e3, min(rhs) * 1e3, {k: round(v[0], 3) for k, v in bs.profile().items() if v[1]}))
    bs.close()

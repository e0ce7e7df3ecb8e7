"""Developer tool (GPU box): record-based right-hand-side re-solve of large-block shapes against the full solve."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rslqr_amd
for (n, m, N, batch) in ((64, 16, 512, 256), (96, 16, 256, 64), (20, 20, 256, 256)):
    bs = rslqr_amd.BatchSolver(n, m, N, batch, flags=rslqr_amd.FLAG_KEEP_RECORDS)
    bs.initialize_synthetic(1)
    bs.solve()
    full = []
    for _ in range(4):
        t0 = time.perf_counter(); bs.solve(); full.append(time.perf_counter() - t0)
    for _ in range(2): bs.solve_rhs_only()
    rhs = []
    for _ in range(6):
        t0 = time.perf_counter(); bs.solve_rhs_only(); rhs.append(time.perf_counter() - t0)
    bs.set_flags(rslqr_amd.FLAG_KEEP_RECORDS | rslqr_amd.FLAG_PROFILE)
    bs.solve()
    bs.profile_reset(); bs.solve_rhs_only()
    print((n, m, N, batch), bs.schedule(), "factor+solve (records kept) %.3f ms   rhs-only %.3f ms" % (min(full) * 1e3, min(rhs) * 1e3),
          {k: round(v[0], 3) for k, v in bs.profile().items() if v[1]})
    bs.close()

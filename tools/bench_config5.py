#!/usr/bin/env python3
"""Times the large-block shape of BASELINE.json config 5, (nx=64, nu=16, N=512), on a small batch
(default 16): generic separator kernel + MFMA Schur kernel. Not the headline bench line."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rslqr_amd as R  # noqa: E402

batch = int(sys.argv[1]) if len(sys.argv) > 1 else 16
bs = R.BatchSolver(64, 16, 512, batch, flags=R.FLAG_PROFILE)
bs.initialize_synthetic(1)
bs.solve()
bs.profile_reset()
t = time.perf_counter()
for _ in range(2):
    bs.solve_async()
bs.synchronize()
dt = (time.perf_counter() - t) / 2
print("(64,16,512)x%d: ms/batch %.3f solves/s %.1f" % (batch, dt * 1e3, batch / dt),
      {k: round(v[0] / 2, 3) for k, v in bs.profile().items() if v[1]})

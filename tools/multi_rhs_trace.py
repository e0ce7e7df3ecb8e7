import os, sys
os.environ["NDLQR_TREE"] = "0"
sys.path.insert(0, "/root/repo")
import numpy as np
import rslqr_amd as R
n, m, N = 12, 4, 256
rng = np.random.default_rng(1)
for batch, nrhs in ((1, 1024), (1024, 1)):
    bs = R.BatchSolver(n, m, N, batch, flags=R.FLAG_KEEP_RECORDS)
    bs.initialize_synthetic(1); bs.solve()
    q, d = rng.standard_normal((nrhs, batch, N, n)), 0.1 * rng.standard_normal((nrhs, batch, N, n))
    r, x0 = rng.standard_normal((nrhs, batch, N, m)), rng.standard_normal((nrhs, batch, n))
    for _ in range(6):
        bs.solve_multi_rhs(q, r, d, x0)
    bs.close()

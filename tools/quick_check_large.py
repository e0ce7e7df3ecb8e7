"""Developer tool (GPU box): default fast path of large-block shapes against the oracle, one line each."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import rslqr_amd as R  # noqa: E402
from support import Oracle, Problem  # noqa: E402

orc = Oracle()
shapes = [(16, 4, 2, 2), (16, 4, 4, 2), (16, 4, 8, 3), (16, 16, 128, 3), (32, 8, 64, 2), (48, 16, 16, 2), (64, 16, 32, 2),
          (64, 16, 512, 2),
          # blocks that do not fill their tiles (zero-padded in LDS)
          (20, 20, 16, 2), (7, 9, 16, 3), (3, 1, 2, 2), (1, 1, 4, 2), (5, 3, 32, 3), (24, 6, 32, 2), (17, 3, 16, 2),
          (33, 5, 8, 2), (50, 10, 64, 2), (64, 15, 16, 2), (63, 1, 16, 2), (15, 2, 256, 2),
          # beyond 64 states: five / six tile columns
          (72, 8, 8, 1), (80, 16, 16, 2), (96, 16, 32, 2), (90, 6, 8, 1), (65, 3, 4, 1), (96, 32, 4, 1),
          (112, 16, 8, 1), (128, 16, 16, 2), (128, 8, 4, 1), (100, 4, 4, 1), (120, 10, 8, 1), (97, 3, 4, 1), (113, 7, 4, 1),
          # inputs wider than a workgroup: the knot-based runtime-sized kernels
          (16, 300, 4, 1)]
for (n, m, N, batch) in shapes:
    bs = R.BatchSolver(n, m, N, batch)
    bs.initialize_synthetic(11)
    worst = 0
    for rep in range(3):  # both buffer sets of the pipeline, and a replay
        rc = bs.solve()
        sol = bs.solutions()
        for p in range(batch):
            g = R.generate_synthetic(n, m, N, 11 + p)
            prob = Problem(n, m, N, g["A"], g["B"], g["Q"], g["R"], g["q"], g["r"], g["d"], g["x0"])
            ref = orc.solve(prob, 1)[0][: prob.nvars]
            worst = max(worst, np.linalg.norm(sol[p] - ref) / np.linalg.norm(ref))
    res, bn = bs.kkt_residuals()
    print((n, m, N, batch), bs.schedule(), "rc", rc, "rel err %.2e" % worst, "kkt %.2e" % (res / np.maximum(1, bn)).max(),
          flush=True)
    bs.close()

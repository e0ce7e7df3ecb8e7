"""Developer tool (GPU box): default fast path of a few shapes against the oracle, one line each.
NDLQR_TREE=0 is forced so that small batches also run the level-per-launch schedule."""
import os
import sys

import numpy as np

os.environ.setdefault("NDLQR_TREE", "0")
os.environ.setdefault("NDLQR_RB_BACKSUB", "1")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import rslqr_amd as R  # noqa: E402
from support import Oracle, Problem  # noqa: E402

orc = Oracle()
for (n, m, N, batch) in [(12, 4, 16, 3), (12, 4, 64, 5), (12, 4, 256, 9), (6, 3, 64, 5), (8, 4, 32, 4), (10, 4, 128, 2),
                         (9, 3, 64, 3), (12, 4, 1024, 2)]:
    bs = R.BatchSolver(n, m, N, batch)
    bs.initialize_synthetic(11)
    rc = bs.solve()
    sol = bs.solutions()
    worst = 0
    for p in range(batch):
        g = R.generate_synthetic(n, m, N, 11 + p)
        prob = Problem(n, m, N, g["A"], g["B"], g["Q"], g["R"], g["q"], g["r"], g["d"], g["x0"])
        ref = orc.solve(prob, 1)[0][: prob.nvars]
        worst = max(worst, np.linalg.norm(sol[p] - ref) / np.linalg.norm(ref))
    res, bn = bs.kkt_residuals()
    print((n, m, N, batch), bs.schedule(), "rc", rc, "rel err %.2e" % worst, "kkt %.2e" % (res / np.maximum(1, bn)).max(),
          flush=True)
    bs.close()

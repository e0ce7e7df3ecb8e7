"""Developer tool (GPU box): the co-scheduled form (NDLQR_COSCHED=1: bottom levels of a solve + back-substitution of the
previous one in one launch) against the oracle -- runs of consecutive solves, new inputs between runs -- and its step
time against the default on the same box.   python tools/cosched_probe.py [--time-only]"""
import os
import sys
import time

import numpy as np

os.environ.setdefault("NDLQR_TREE", "0")
os.environ["NDLQR_COSCHED"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import rslqr_amd as R  # noqa: E402
from support import Oracle, Problem  # noqa: E402


def worst_error(orc, bs, n, m, N, batch, seed):
    sol = bs.solutions()
    worst = 0
    for p in range(batch):
        g = R.generate_synthetic(n, m, N, seed + p)
        prob = Problem(n, m, N, g["A"], g["B"], g["Q"], g["R"], g["q"], g["r"], g["d"], g["x0"])
        ref = orc.solve(prob, 1)[0][: prob.nvars]
        worst = max(worst, np.linalg.norm(sol[p] - ref) / np.linalg.norm(ref))
    return worst


if "--time-only" not in sys.argv:
    orc = Oracle()
    for (n, m, N, batch) in [(12, 4, 16, 3), (12, 4, 64, 5), (12, 4, 256, 9), (10, 4, 128, 2), (9, 3, 64, 3),
                             (13, 4, 64, 3), (12, 4, 1024, 2), (11, 3, 64, 4)]:
        bs = R.BatchSolver(n, m, N, batch)
        out = []
        for run, (seed, nsolves) in enumerate([(11, 1), (23, 2), (37, 5), (41, 4)]):
            bs.initialize_synthetic(seed)
            for _ in range(nsolves):
                bs.solve_async()
            rc = bs.synchronize()
            res, bn = bs.kkt_residuals()
            out.append("%dx rc %d err %.1e kkt %.1e" % (nsolves, rc, worst_error(orc, bs, n, m, N, batch, seed),
                                                        (res / np.maximum(1, bn)).max()))
        print((n, m, N, batch), bs.schedule(), " | ".join(out), flush=True)
        bs.close()

shapes = [(12, 4, 256, 1024), (12, 4, 1024, 512), (13, 4, 256, 1024), (9, 3, 256, 1024), (10, 4, 256, 1024)]
if os.environ.get("NDLQR_PROBE_SHAPES"):  # "n,m,N,batch;..."
    shapes = [tuple(int(v) for v in sh.split(",")) for sh in os.environ["NDLQR_PROBE_SHAPES"].split(";")]
modes = tuple(os.environ.get("NDLQR_PROBE_COSCHED", "0,1").split(","))
for (n, m, N, batch) in shapes:
    line = []
    for cos in modes:
        os.environ["NDLQR_COSCHED"] = cos
        bs = R.BatchSolver(n, m, N, batch)
        bs.initialize_synthetic(1)
        for _ in range(40):
            bs.solve_async()
        bs.synchronize()
        best = 1e9
        for rep in range(5):
            t0 = time.perf_counter()
            for _ in range(50):
                bs.solve_async()
            bs.synchronize()
            best = min(best, (time.perf_counter() - t0) / 50 * 1e3)
        res, bn = bs.kkt_residuals()
        line.append("cosched=%s %s %.4f ms/step kkt %.1e" % (cos, bs.schedule(), best, (res / np.maximum(1, bn)).max()))
        bs.close()
    print((n, m, N, batch), " | ".join(line), flush=True)

"""Developer tool (GPU box): how far do the fast paths follow the oracle into ill-conditioning? Error of the default fast mode
and of the oracle against the refined (extended-precision) solution over harsher families than the tests use."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import support, rslqr_amd as R

o = support.Oracle()
fams = [(1.0, 1.0, 1.0), (1.0, 1.0, 1e-4), (1.0, 1.0, 1e-6), (1.0, 1.0, 1e-8), (1.0, 1e-6, 1.0), (1.0, 1e-6, 1e-6),
        (1.3, 1e-3, 1.0), (1.5, 1.0, 1.0), (1.5, 1e-4, 1e-4), (0.5, 1e3, 1e3)]
for (n, m, N, batch) in [(12, 4, 256, 40), (6, 3, 256, 64), (64, 16, 64, 1), (50, 10, 32, 1), (4, 2, 128, 2)]:
    print("shape (%d,%d,%d) x %d" % (n, m, N, batch))
    for fa, fq, fr in fams:
        gs = []
        for p in range(batch):
            g = R.generate_synthetic(n, m, N, 11 + p)
            g["A"] *= fa; g["Q"] *= fq; g["R"] *= fr
            gs.append(g)
        prob = support.Problem(n, m, N, *[gs[0][k] for k in ("A", "B", "Q", "R", "q", "r", "d", "x0")])
        try:
            truth = support.refined_solution(o, prob, 4)
        except Exception as e:
            print("  fam", (fa, fq, fr), "refinement failed", e); continue
        zo = o.solve(prob, 8)[0][: prob.nvars]
        bs = R.BatchSolver(n, m, N, batch)
        bs.initialize_flat(*[np.stack([g[k] for g in gs]) for k in ("A", "B", "Q", "R", "q", "r", "d", "x0")])
        rc = bs.solve()
        z = bs.solutions()[0]
        sched = bs.schedule()
        bs.close()
        nt = np.linalg.norm(truth)
        eo, eg = np.linalg.norm(zo - truth) / nt, np.linalg.norm(z - truth) / nt
        print("  A x %-4g Q x %-6g R x %-6g  oracle %.1e  gpu %.1e  ratio %6.1f  rc %d [%s]%s" % (
            fa, fq, fr, eo, eg, eg / max(eo, 1e-17), rc, sched, "   <-- " if eg > 10 * eo and eg > 1e-9 else ""))

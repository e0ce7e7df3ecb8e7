"""Developer tool (GPU box): one fuzz case -- errors of the oracle and of the GPU paths against the refined
(extended-precision) solution.   python tools/fuzz_diag.py n m N batch seed a_scale q_scale r_scale [problem]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import support, rslqr_amd as R

n, m, N, batch, seed = (int(x) for x in sys.argv[1:6])
fa, fq, fr = (float(x) for x in sys.argv[6:9])
pp = int(sys.argv[9]) if len(sys.argv) > 9 else 0
o = support.Oracle()
gens = [R.generate_synthetic(n, m, N, seed + p) for p in range(batch)]
for g in gens:
    g["A"] = g["A"] * fa; g["Q"] = g["Q"] * fq; g["R"] = g["R"] * fr
probs = [support.Problem(n, m, N, *[g[k] for k in ("A", "B", "Q", "R", "q", "r", "d", "x0")]) for g in gens]
truth = support.refined_solution(o, probs[pp], 3)
zo = o.solve(probs[pp], 8)[0][: probs[pp].nvars]
rel = lambda z: np.linalg.norm(z - truth) / np.linalg.norm(truth)  # noqa: E731
print("oracle vs refined %.2e   (oracle KKT %.2e)" % (rel(zo), (lambda r: r[0] / max(1, r[1]))(o.kkt_residual(probs[pp], zo))))
for name, flags, env in (("default", 0, {}), ("generic-flag", R.FLAG_GENERIC, {}), ("keep_fact", R.FLAG_KEEP_FACT, {}),
                         ("strict", R.FLAG_STRICT_FP, {}), ("no-pad", 0, {"NDLQR_NO_PAD": "1"})):
    os.environ.update(env)
    bs = R.BatchSolver(n, m, N, batch, flags=flags)
    for k in env:
        del os.environ[k]
    bs.initialize_flat(*[np.stack([getattr(p, k) for p in probs]) for k in ("A", "B", "Q", "R", "q", "r", "d", "x0")])
    rc = bs.solve()
    z = bs.solutions()[pp]
    res, bn = o.kkt_residual(probs[pp], z)
    print("%-14s [%s] rc %d vs refined %.2e  vs oracle %.2e  KKT %.2e" % (name, bs.schedule(), rc, rel(z),
          np.linalg.norm(z - zo) / np.linalg.norm(zo), res / max(1, bn)), flush=True)
    bs.close()

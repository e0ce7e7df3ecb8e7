"""Developer tool (GPU box): where does the fast path lose accuracy on an ill-conditioned family?
    python tools/hard_diag.py [n m N batch a_scale q_scale r_scale]
Prints the error of every schedule against the refined (extended-precision) solution, split by variable kind and by
the tree level of the knot's separator."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import support, rslqr_amd as R

a = sys.argv[1:]
n, m, N, batch = (int(x) for x in a[:4]) if len(a) >= 4 else (12, 4, 256, 40)
fa, fq, fr = (float(x) for x in a[4:7]) if len(a) >= 7 else (1.0, 1.0, 1e-4)
o = support.Oracle()
gs, probs = [], []
for p in range(batch):
    g = R.generate_synthetic(n, m, N, 11 + p)
    g["A"] *= fa; g["Q"] *= fq; g["R"] *= fr
    gs.append(g); probs.append(support.Problem(n, m, N, *[g[k] for k in ("A", "B", "Q", "R", "q", "r", "d", "x0")]))
truth = support.refined_solution(o, probs[0], 3)
zo = o.solve(probs[0], 8)[0][: probs[0].nvars]
zb = 2 * n + m
lvl = np.array([(lambda k: (k ^ (k + 1)).bit_length() - 1)(k) for k in range(N)])  # level of separator k (trailing ones)


def report(name, z):
    full = np.zeros(N * zb); full[: z.size] = z
    t = np.zeros(N * zb); t[: truth.size] = truth
    E = (full - t).reshape(N, zb); T = t.reshape(N, zb)
    parts = {"lam": slice(0, n), "x": slice(n, 2 * n), "u": slice(2 * n, zb)}
    s = "%-26s total %.2e |" % (name, np.linalg.norm(E) / np.linalg.norm(T))
    for k, sl in parts.items():
        s += " %s %.2e" % (k, np.linalg.norm(E[:, sl]) / np.linalg.norm(T[:, sl]))
    s += " | y by sep level:"
    for l in range(int(np.log2(N))):
        rows = np.where(lvl[: N - 1] == l)[0] + 1  # lambda of knot k+1 = multiplier of separator k
        s += " %d:%.1e" % (l, np.linalg.norm(E[rows, :n]) / max(np.linalg.norm(T[rows, :n]), 1e-300))
    print(s, flush=True)


report("oracle", zo)
def run(name, b, flags=0, env=None):
    for k, v in (env or {}).items():
        os.environ[k] = v
    bs = R.BatchSolver(n, m, N, b, flags=flags)
    for k in (env or {}):
        del os.environ[k]
    bs.initialize_flat(*[np.stack([getattr(p, k) for p in probs[:b]]) for k in ("A", "B", "Q", "R", "q", "r", "d", "x0")])
    assert bs.solve() == 0
    report(name + " [" + bs.schedule() + "]", bs.solutions()[0])
    bs.close()
run("default batch", batch)
run("default batch 1", 1)
run("rowbcast=1", batch, env={"NDLQR_ROWBCAST": "1"})
run("rowbcast=0", batch, env={"NDLQR_ROWBCAST": "0"})
run("keep_records", batch, flags=R.FLAG_KEEP_RECORDS)
run("keep_fact (knot)", 1, flags=R.FLAG_KEEP_FACT)
run("generic", 1, flags=R.FLAG_GENERIC)
run("no_top", batch, env={"NDLQR_NO_TOP": "1"})

"""Developer tool (CPU): fp64 numpy model of the separator-only ("reduced") schedule of DESIGN.md section 3.1, to
study where the fast path loses accuracy on ill-conditioned families before touching a kernel.
    python tools/reduced_model.py [n m N a_scale q_scale r_scale]
Variants: how X = S-bar^-1 R is formed (explicit inverse W'W / substitutions) and what the back-substitution uses
(records X / factor L). Error of lambda, x, u against the refined (extended-precision) solution."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import support, rslqr_amd as R


def chol(M):
    k = M.shape[0]; L = np.zeros_like(M)
    for j in range(k):
        L[j, j] = np.sqrt(M[j, j] - L[j, :j] @ L[j, :j])
        for i in range(j + 1, k):
            L[i, j] = (M[i, j] - L[i, :j] @ L[j, :j]) / L[j, j]
    return L


def fsub(L, Bm):
    Y = np.zeros_like(Bm)
    for i in range(L.shape[0]):
        Y[i] = (Bm[i] - L[i, :i] @ Y[:i]) / L[i, i]
    return Y


def bsub(L, Bm):
    Y = np.zeros_like(Bm)
    for i in reversed(range(L.shape[0])):
        Y[i] = (Bm[i] - L[i + 1:, i] @ Y[i + 1:]) / L[i, i]
    return Y


def solve_reduced(p, form="inverse", backsub="records"):
    n, m, N = p.n, p.m, p.N
    A = p.A.reshape(N, n, n).transpose(0, 2, 1); B = p.B.reshape(N, m, n).transpose(0, 2, 1)
    Q, Rr, q, r, d, x0 = p.Q, p.R, p.q, p.r, p.d, p.x0
    M = N - 1
    S = [None] * M; ra = [None] * M; rb = [None] * M; bt = [None] * M
    for s in range(M):
        AQ = A[s] / Q[s]; BR = B[s] / Rr[s]
        if s == 0:
            S[s] = BR @ B[s].T + np.diag(1 / Q[1]); ra[s] = np.zeros((n, n))
            bt[s] = d[0] + A[0] @ x0 - BR @ r[0] + q[1] / Q[1]
        else:
            S[s] = AQ @ A[s].T + BR @ B[s].T + np.diag(1 / Q[s + 1]); ra[s] = -AQ
            bt[s] = d[s] - AQ @ q[s] - BR @ r[s] + q[s + 1] / Q[s + 1]
        rb[s] = -(A[s + 1] / Q[s + 1]).T if s + 1 < M else np.zeros((n, n))   # -Q^-1 A'
        if s + 1 < M:
            rb[s] = -(A[s + 1].T / Q[s + 1][:, None])
    rec = [None] * M
    K = int(np.log2(N))
    for l in range(K):
        step = 1 << l
        for s in range(step - 1, M, 2 * step):
            a_, b_ = s - step, s + step
            L = chol(S[s])
            Rp = np.concatenate([ra[s], bt[s][:, None], rb[s]], axis=1)
            if form == "wy":   # Y by substitution, X = W'Y, pushes Y'Y (chol_wy_mc)
                W = fsub(L, np.eye(n)); Y = fsub(L, Rp); X = W.T @ Y
            elif form == "two+gram":
                W = fsub(L, np.eye(n)); Y = W @ Rp; X = W.T @ Y
            elif form == "inverse":
                W = fsub(L, np.eye(n)); X = (W.T @ W) @ Rp
            elif form == "two":
                W = fsub(L, np.eye(n)); X = W.T @ (W @ Rp)
            else:
                X = bsub(L, fsub(L, Rp))
            fa, z, fb = X[:, :n], X[:, n], X[:, n + 1:]
            rec[s] = (fa, z, fb, L, ra[s].copy(), rb[s].copy(), bt[s].copy(), a_, b_)
            if form in ("gram", "two+gram", "wy"):   # pushes as Y'Y, Y = L^-1 R
                if form == "gram":
                    Y = fsub(L, Rp)
                Ya, yz, Yb = Y[:, :n], Y[:, n], Y[:, n + 1:]
                gaa, gab, gbb, gaz, gbz = Ya.T @ Ya, Ya.T @ Yb, Yb.T @ Yb, Ya.T @ yz, Yb.T @ yz
            else:
                gaa, gab, gbb, gaz, gbz = ra[s].T @ fa, ra[s].T @ fb, rb[s].T @ fb, ra[s].T @ z, rb[s].T @ z
            if a_ >= 0:
                S[a_] = S[a_] - gaa; bt[a_] = bt[a_] - gaz
                rb[a_] = -gab if b_ < M else np.zeros((n, n))
            if b_ < M:
                S[b_] = S[b_] - gbb; bt[b_] = bt[b_] - gbz
                ra[b_] = -gab.T if a_ >= 0 else np.zeros((n, n))
    y = np.zeros((M, n))
    for l in reversed(range(K)):
        step = 1 << l
        for s in range(step - 1, M, 2 * step):
            fa, z, fb, L, ra_s, rb_s, bt_s, a_, b_ = rec[s]
            yA = y[a_] if a_ >= 0 else np.zeros(n); yB = y[b_] if b_ < M else np.zeros(n)
            if backsub == "records" or (backsub in ("inv0", "fac0") and l > 0):
                y[s] = z - fa @ yA - fb @ yB
            elif backsub == "inv0":  # level 0: the explicit inverse applied to the re-formed right-hand side (rb_bottom)
                W = fsub(L, np.eye(n))
                y[s] = (W.T @ W) @ (bt_s - ra_s @ yA - rb_s @ yB)
            else:
                y[s] = bsub(L, fsub(L, (bt_s - ra_s @ yA - rb_s @ yB)[:, None]))[:, 0]
    zb = 2 * n + m
    out = np.zeros((N, zb))
    for k in range(N):
        ykm = y[k - 1] if k > 0 else None
        yk = y[k] if k < M else np.zeros(n)
        if k == 0:
            out[0, n:2 * n] = x0; out[0, :n] = Q[0] * x0 + q[0] + A[0].T @ yk
        else:
            out[k, :n] = ykm; out[k, n:2 * n] = (-q[k] - (A[k].T @ yk if k < M else 0) + ykm) / Q[k]
        if k < M:
            out[k, 2 * n:] = (-r[k] - B[k].T @ yk) / Rr[k]
    return out.reshape(-1)[: p.nvars]


if __name__ == "__main__":
    a = sys.argv[1:]
    n, m, N = (int(x) for x in a[:3]) if len(a) >= 3 else (12, 4, 64)
    fa_, fq, fr = (float(x) for x in a[3:6]) if len(a) >= 6 else (1.0, 1.0, 1e-4)
    g = R.generate_synthetic(n, m, N, 11)
    g["A"] *= fa_; g["Q"] *= fq; g["R"] *= fr
    p = support.Problem(n, m, N, *[g[k] for k in ("A", "B", "Q", "R", "q", "r", "d", "x0")])
    o = support.Oracle()
    truth = support.refined_solution(o, p, 3)
    zb = 2 * n + m
    def rep(name, z):
        f = np.zeros(N * zb); f[: z.size] = z; t = np.zeros(N * zb); t[: truth.size] = truth
        E = (f - t).reshape(N, zb); T = t.reshape(N, zb)
        print("%-28s total %.2e lam %.2e x %.2e u %.2e" % (name, np.linalg.norm(E) / np.linalg.norm(T),
              np.linalg.norm(E[:, :n]) / np.linalg.norm(T[:, :n]), np.linalg.norm(E[:, n:2*n]) / np.linalg.norm(T[:, n:2*n]),
              np.linalg.norm(E[:, 2*n:]) / np.linalg.norm(T[:, 2*n:])))
    rep("oracle", o.solve(p, 8)[0][: p.nvars])
    for form in ("inverse", "two", "subst", "gram", "two+gram", "wy"):
        for bs in ("records", "factor", "inv0", "fac0"):
            rep(form + " / " + bs, solve_reduced(p, form, bs))
    print("|y| %.2e |u| %.2e" % (np.abs(truth.reshape(-1)[:0]).sum(), 0))

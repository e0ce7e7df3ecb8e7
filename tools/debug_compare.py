"""Developer tool (GPU box): separator records and accumulator slots of the row-broadcast schedule
against the matrix-core schedule on the same problem, element by element (first mismatches)."""
import ctypes as C
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def run(n, m, N, batch, seed):
    import rslqr_amd as R
    bs = R.BatchSolver(n, m, N, batch)
    bs.initialize_synthetic(seed)
    rc = bs.solve()
    L = R.lib()
    L.ndlqr_hip_debug_download.restype = C.c_long
    L.ndlqr_hip_debug_download.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_double), C.c_long]
    rec = np.zeros(batch * N * (2 * n * n + n))
    slot = (4 * n * n + 2 * n + 15) // 16 * 16
    red = np.zeros(batch * (N // 4) * slot)
    L.ndlqr_hip_debug_download(bs.ctx, 0, rec.ctypes.data_as(C.POINTER(C.c_double)), rec.size)
    L.ndlqr_hip_debug_download(bs.ctx, 1, red.ctypes.data_as(C.POINTER(C.c_double)), red.size)
    return rc, bs.schedule(), rec.reshape(batch, N, -1), red.reshape(batch, N // 4, slot), bs.solutions()


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "child":
        n, m, N, batch, seed = [int(v) for v in sys.argv[2:7]]
        rc, sched, rec, red, sol = run(n, m, N, batch, seed)
        np.savez(sys.argv[7], rc=rc, sched=sched, rec=rec, red=red, sol=sol)
        sys.exit(0)
    n, m, N, batch, seed = [int(v) for v in (sys.argv[1:6] if len(sys.argv) > 5 else (8, 4, 16, 1, 11))]
    out = {}
    for tag, env in (("rb", {"NDLQR_TREE": "0"}), ("mc", {"NDLQR_TREE": "0", "NDLQR_ROWBCAST": "0"})):
        e = dict(os.environ); e.update(env)
        f = "/tmp/dbg_%s.npz" % tag
        subprocess.run([sys.executable, __file__, "child", str(n), str(m), str(N), str(batch), str(seed), f], env=e, check=True)
        out[tag] = np.load(f)
    import rslqr_amd as R
    g = R.generate_synthetic(n, m, N, seed)
    def sinv(sidx):
        A = g["A"][sidx].reshape(n, n, order="F"); B = g["B"][sidx].reshape(n, m, order="F")
        S = B @ np.diag(1 / g["R"][sidx]) @ B.T + np.diag(1 / g["Q"][sidx + 1])
        if sidx > 0:
            S = S + A @ np.diag(1 / g["Q"][sidx]) @ A.T
        return np.linalg.inv(S)
    print("schedules", out["rb"]["sched"], out["mc"]["sched"], "rc", out["rb"]["rc"], out["mc"]["rc"])
    NN = n * n
    for s in range(N - 1):
        lvl = 0
        t = s
        while t & 1:
            lvl += 1; t >>= 1
        a, b_ = out["rb"]["rec"][0, s], out["mc"]["rec"][0, s]
        if lvl == 0:
            Si = sinv(s)
            packed = np.array([Si[i, c] for i in range(n) for c in range(i + 1)])
            print("sep %3d lvl0: rb S^-1 packed finite=%s  max abs err vs numpy %.3e" % (s, np.isfinite(a[: n * (n + 1) // 2]).all(), np.abs(a[:packed.size] - packed).max()))
        else:
            err = np.abs(a - b_).max()
            print("sep %3d lvl%d: rec max abs diff %.3e (|mc| max %.3e) finite=%s" % (s, lvl, err, np.abs(b_).max(), np.isfinite(a).all()))
    for q in range(N // 4):
        a, b_ = out["rb"]["red"][0, q], out["mc"]["red"][0, q]
        names = ["DL", "DR", "CA", "CB"]
        msg = []
        for k, nm in enumerate(names):
            msg.append("%s %.2e" % (nm, np.abs(a[k * NN:(k + 1) * NN] - b_[k * NN:(k + 1) * NN]).max()))
        msg.append("gL %.2e" % np.abs(a[4 * NN:4 * NN + n] - b_[4 * NN:4 * NN + n]).max())
        msg.append("gR %.2e" % np.abs(a[4 * NN + n:4 * NN + 2 * n] - b_[4 * NN + n:4 * NN + 2 * n]).max())
        print("slot of sep %3d: " % (4 * q + 3) + "  ".join(msg))
    print("solution max abs diff", np.abs(out["rb"]["sol"] - out["mc"]["sol"]).max())

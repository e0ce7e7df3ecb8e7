"""Developer tool (GPU box): random shapes, batches, flags and call sequences against the oracle.
    python tools/fuzz_parity.py [seconds] [seed]
Every case: full solve (twice: both buffer sets), optionally a right-hand-side re-solve / an MPC step / multiple
right-hand sides, solutions against the oracle (bit-exact in strict mode, <= 1e-9 relative otherwise; on draws so
ill-conditioned that the oracle's own KKT residual exceeds 1e-11: a KKT residual within 10x the oracle's)."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import rslqr_amd as R  # noqa: E402
from support import Oracle, Problem  # noqa: E402

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
orc = Oracle()
INST = [(12, 4), (6, 3), (13, 4), (8, 4), (4, 2), (10, 4), (9, 3), (5, 2), (4, 1), (2, 1), (8, 16), (12, 8), (15, 2)]
t_end = time.time() + budget
cases = fails = 0
t_note = time.time()
while time.time() < t_end:
    if time.time() - t_note > 45:
        t_note = time.time()
        print("progress: %d cases, %d failures" % (cases, fails), flush=True)
    kind = rng.integers(0, 10)
    if kind < 5:
        n, m = INST[rng.integers(0, len(INST))]
    elif kind < 8:
        n, m = int(rng.integers(1, 40)), int(rng.integers(1, 12))
    else:
        n, m = int(rng.integers(40, 150)), int(rng.integers(1, 20))
    N = int(2 ** rng.integers(1, 9 if n < 40 else 5))
    batch = int(rng.integers(1, 40 if n < 40 else 4))
    flags = [0, 0, 0, R.FLAG_STRICT_FP, R.FLAG_KEEP_FACT, R.FLAG_KEEP_RECORDS, R.FLAG_STRICT_FP | R.FLAG_KEEP_FACT,
             R.FLAG_GENERIC][rng.integers(0, 8)]
    tree = ["0", "1", None][rng.integers(0, 3)]
    if tree is None:
        os.environ.pop("NDLQR_TREE", None)
    else:
        os.environ["NDLQR_TREE"] = tree
    seed = int(rng.integers(1, 1 << 30))
    desc = "(%d,%d,%d)x%d flags %d tree %s seed %d" % (n, m, N, batch, flags, tree, seed)
    try:
        gens = [R.generate_synthetic(n, m, N, seed + p) for p in range(batch)]
        fam = [(1.0, 1.0, 1.0)] * 3 + [(1.15, 1.0, 1.0), (1.0, 1e-4, 1.0), (1.3, 1e-3, 1.0), (1.0, 1.0, 1e-4)]
        a_s, q_s, r_s = fam[rng.integers(0, len(fam))]  # (the ill-conditioned families of tests/test_gpu_parity.py)
        if N > 64 and a_s > 1.0:
            a_s = 1.0  # (unstable dynamics over long horizons: the oracle itself loses digits)
        for g in gens:
            g["A"] = g["A"] * a_s; g["Q"] = g["Q"] * q_s; g["R"] = g["R"] * r_s
        desc += " fam (%g,%g,%g)" % (a_s, q_s, r_s)
        probs = [Problem(n, m, N, g["A"], g["B"], g["Q"], g["R"], g["q"], g["r"], g["d"], g["x0"]) for g in gens]
        flat = [np.stack([g[k] for g in gens]) for k in ("A", "B", "Q", "R", "q", "r", "d", "x0")]
        bs = R.BatchSolver(n, m, N, batch, flags=flags)
        bs.initialize_flat(*flat)
        check = sorted(set([0, batch - 1]))

        def compare(sol, ps, what):
            global fails
            for p in check:
                ref = orc.solve(ps[p], 4)[0][: ps[p].nvars]
                if flags & R.FLAG_STRICT_FP:
                    ok = np.array_equal(sol[p], ref)
                else:
                    ok = np.linalg.norm(sol[p] - ref) / np.linalg.norm(ref) <= 1e-9
                    if not ok:
                        # an ill-conditioned draw (unstable dynamics, one weak input): the oracle's own solution is off
                        # by more than the tolerance -- then the bar is the KKT residual, within 10x the oracle's (the bar
                        # of tests/test_gpu_parity.py::test_harder_families_*)
                        ores, obn = orc.kkt_residual(ps[p], ref)
                        res, bn = orc.kkt_residual(ps[p], sol[p])
                        if ores / max(1.0, obn) > 1e-11 and res / max(1.0, bn) <= 10.0 * ores / max(1.0, obn):
                            ok = True
                            print("note: ill-conditioned draw", desc, what, "oracle KKT %.1e, device KKT %.1e" % (
                                ores / max(1.0, obn), res / max(1.0, bn)), flush=True)
                if not ok:
                    fails += 1
                    print("MISMATCH", desc, what, "problem", p, bs.schedule(),
                          np.linalg.norm(sol[p] - ref) / np.linalg.norm(ref), flush=True)
        assert bs.solve() == 0
        compare(bs.solutions(), probs, "solve")
        bs.solve_async(); bs.solve_async()
        assert bs.synchronize() == 0
        compare(bs.solutions(), probs, "solve x3")
        other = [R.generate_synthetic(n, m, N, seed + 1000 + p) for p in range(batch)]
        mixed = [Problem(n, m, N, a.A, a.B, a.Q, a.R, o["q"], o["r"], o["d"], o["x0"]) for a, o in zip(probs, other)]
        oflat = [np.stack([o[k] for o in other]) for k in ("q", "r", "d", "x0")]
        if flags & (R.FLAG_KEEP_FACT | R.FLAG_KEEP_RECORDS) and rng.integers(0, 2):
            bs.set_rhs_flat(*oflat)
            rc = bs.solve_rhs_only()
            if rc == 0:
                compare(bs.solutions(), mixed, "rhs-only")
            else:
                print("note: rhs-only refused", desc, bs.schedule(), flush=True)
        if not (flags & R.FLAG_STRICT_FP) and rng.integers(0, 2):
            pins = [R.pinned_empty(a.shape) for a in oflat]
            for pa, a in zip(pins, oflat):
                pa[...] = a
            outs = [R.pinned_empty((batch, bs.nvars)) for _ in range(3)]
            for s in range(3):
                assert bs.step_async(pins[0], pins[1], pins[2], pins[3], outs[s]) == 0
            assert bs.synchronize() == 0
            for s in range(3):
                compare(outs[s], mixed, "step %d" % s)
            if rng.integers(0, 2):
                # a knot range computed alone (NDLQR_SOLN_ONLY), x0 from pinned or device memory, the slice into either
                k0 = int(rng.integers(0, N)); nk = int(rng.integers(1, min(N - k0, 12) + 1))
                blocks = int(rng.integers(1, 8))
                bs.set_step_selection(k0, nk, blocks | R.SOLN_ONLY)
                width = bs.slice_width(blocks)
                xsrc = R.DeviceArray(pins[3].shape).set(pins[3]) if rng.integers(0, 2) else pins[3]
                dsts = [R.DeviceArray((batch, nk, width)) if rng.integers(0, 2) else R.pinned_empty((batch, nk, width)) for _ in range(2)]
                for dst in dsts:
                    assert bs.step_async(None, None, None, xsrc, dst) == 0
                assert bs.synchronize() == 0
                zb = 2 * n + m
                cols = ([*range(0, n)] if blocks & 1 else []) + ([*range(n, 2 * n)] if blocks & 2 else []) + \
                       ([*range(2 * n, zb)] if blocks & 4 else [])
                for dst in dsts:
                    got = dst.get() if isinstance(dst, R.DeviceArray) else dst
                    for pp in check:
                        full = np.zeros(N * zb); full[: mixed[pp].nvars] = outs[2][pp]
                        want = full.reshape(N, zb)[k0:k0 + nk][:, cols]
                        # (against the full step of the same right-hand side: equal to rounding -- bit for bit unless the
                        #  records are kept, where the full step was the factorisation and these are re-solves)
                        if not np.allclose(got[pp], want, rtol=0, atol=1e-11 * max(1.0, np.abs(outs[2][pp]).max())):
                            fails += 1
                            print("MISMATCH", desc, "selection (%d,%d,%d) computed alone" % (k0, nk, blocks), pp, bs.schedule(),
                                  np.abs(got[pp] - want).max(), flush=True)
                bs.set_step_selection()
        if flags == R.FLAG_KEEP_RECORDS and bs.schedule().startswith("reduced-compact-records") and rng.integers(0, 2):
            nrhs = int(rng.integers(1, 4))
            q4 = np.stack([oflat[0]] * nrhs); r4 = np.stack([oflat[1]] * nrhs)
            d4 = np.stack([oflat[2]] * nrhs); x4 = np.stack([oflat[3]] * nrhs)
            sol = bs.solve_multi_rhs(q4, r4, d4, x4)
            compare(sol[nrhs - 1], mixed, "multi-rhs")
        bs.close()
        cases += 1
    except Exception as e:  # noqa: BLE001
        fails += 1
        print("EXCEPTION", desc, repr(e)[:300], flush=True)
print("fuzz: %d cases, %d failures" % (cases, fails), flush=True)
sys.exit(1 if fails else 0)

#!/usr/bin/env python3
"""Generates tests/golden/synthetic_ref.npz from the REAL reference (oracle/_ref/libref.so, built
from /root/reference/src by oracle/Makefile). Run in the build container only:

    make -C oracle && python tests/golden/make_golden.py

The fixture holds, for a few seeded synthetic problems (inputs regenerated at test time from the
seed by ndlqr_GenerateSyntheticFlat, inputs are NOT stored): the reference's solution vector, its
KKT residual, and a few factor blocks after the full solve. Data only -- no reference source.
The JSON files next to this script are the reference's own test fixtures (lqr_prob.json,
lqr_prob_256.json, sample_problem.json, lqr_data.json), copied verbatim as data.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from support import Oracle, Problem, Reference, have_reference, split_block  # noqa: E402

CASES = [  # (n, m, N, seed)
    (12, 4, 16, 1), (12, 4, 64, 2), (12, 4, 256, 3), (6, 3, 32, 4), (6, 3, 256, 5),
    (5, 2, 8, 6), (3, 1, 2, 7), (16, 8, 16, 8), (2, 2, 128, 9),
]


def main():
    assert have_reference(), "build oracle/_ref first (make -C oracle)"
    import rslqr_amd
    ref, orc = Reference(), Oracle()
    out = {"cases": np.array(CASES, dtype=np.int64)}
    for (n, m, N, seed) in CASES:
        g = rslqr_amd.generate_synthetic(n, m, N, seed)
        prob = Problem(n, m, N, g["A"], g["B"], g["Q"], g["R"], g["q"], g["r"], g["d"], g["x0"])
        rs = ref.solver(prob)
        rs.solve(1)
        z = rs.soln()[: prob.nvars].copy()
        res, bnorm = orc.kkt_residual(prob, z)
        key = "n%d_m%d_N%d_s%d" % (n, m, N, seed)
        out[key + "_soln"] = z
        out[key + "_kkt"] = np.array([res, bnorm])
        fact = rs.fact()
        K = prob.K
        # a few blocks: top column of first/middle/last knot, column 0 of knot 1
        picks = [(0, K - 1), (N // 2, K - 1), (N - 1, K - 1), (1, 0)]
        for (k, lvl) in picks:
            lam, st, inp = split_block(fact, prob, k, lvl)
            out[key + "_F%d_%d" % (k, lvl)] = np.concatenate([lam.ravel(), st.ravel(), inp.ravel()])
        # checksum of the whole factor array (sum and sum of squares)
        out[key + "_factsum"] = np.array([fact.sum(), (fact * fact).sum()])
        print(key, "kkt", res, "|soln|", np.linalg.norm(z))
    path = os.path.join(HERE, "synthetic_ref.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()

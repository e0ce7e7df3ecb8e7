"""Oracle vs the real reference compiled from /root/reference/src (oracle/_ref/libref.so).
Only runs where that library exists (the build container); skipped on the GPU box."""
import os

import numpy as np
import pytest

from support import GOLDEN, Problem, Reference, have_reference, load_json_problem

pytestmark = pytest.mark.skipif(not have_reference(), reason="oracle/_ref/libref.so not built")


@pytest.fixture(scope="module")
def ref():
    return Reference()


def synth(ndlqr, n, m, N, seed):
    g = ndlqr.generate_synthetic(n, m, N, seed)
    return Problem(n, m, N, g["A"], g["B"], g["Q"], g["R"], g["q"], g["r"], g["d"], g["x0"])


@pytest.mark.parametrize("fname", ["lqr_prob.json", "lqr_prob_256.json"])
def test_fixture_bit_identical(oracle, ref, fname):
    prob, soln = load_json_problem(os.path.join(GOLDEN, fname))
    z, fact, _, _ = oracle.solve(prob, 1, want_fact=True)
    rs = ref.solver(prob)
    rs.solve(1)
    assert np.array_equal(z, rs.soln())
    assert np.array_equal(fact, rs.fact())
    assert np.linalg.norm(rs.soln()[: prob.nvars] - soln) < 1e-6  # reference vs its own golden


@pytest.mark.parametrize("n,m,N", [(12, 4, 32), (6, 3, 64), (5, 2, 8), (3, 1, 2), (16, 8, 16), (2, 1, 4)])
# nstates = 1 is outside the reference's domain: ndlqr_NewNdData(width=1) forces depth 1 (src/nddata.c:23-29),
# so its matrix factors lose their levels. The oracle and the GPU path handle it; no parity claim there.
def test_synthetic_bit_identical_every_phase(oracle, ref, ndlqr, n, m, N):
    prob = synth(ndlqr, n, m, N, 11)
    o, r = oracle.solver(prob), ref.solver(prob)
    OL, RL = oracle.L, ref.L
    assert np.array_equal(o.data(), r.data()) and np.array_equal(o.soln(), r.soln())  # KKT assembly
    for k in range(N):
        OL.oracle_solve_leaf(o.h, k)
    RL.ndlqr_SolveLeaves(r.h)
    assert np.array_equal(o.fact(), r.fact()) and np.array_equal(o.soln(), r.soln())
    K = prob.K
    tree, rdata, rfact = RL.ref_tree(r.h), RL.ref_data_nd(r.h), RL.ref_fact_nd(r.h)
    for level in range(K):
        nleaf = 1 << (K - level - 1)
        for leaf in range(nleaf):
            idx = (1 << level) * (2 * leaf + 1) - 1
            assert idx == RL.ndlqr_GetIndexFromLeaf(tree, leaf, level)
            for up in range(level, K):
                OL.oracle_inner_product(o.h, 0, idx, level, up)
                RL.ndlqr_FactorInnerProduct(rdata, rfact, idx, level, up)
        assert np.array_equal(o.fact(), r.fact())
        for leaf in range(nleaf):
            assert OL.oracle_factor_separator(o.h, leaf, level) == RL.ref_factor_separator(r.h, leaf, level)
        assert np.array_equal(o.fact(), r.fact())
        for leaf in range(nleaf):
            idx = (1 << level) * (2 * leaf + 1) - 1
            for up in range(level + 1, K):
                OL.oracle_solve_chol_factor(o.h, idx, level, up)
                RL.ref_solve_chol_factor(r.h, leaf, level, up)
        assert np.array_equal(o.fact(), r.fact())
        for k in range(N):
            idx = OL.oracle_index_at_level(k, level)
            assert idx == RL.ndlqr_GetIndexAtLevel(tree, k, level)
            cl = OL.oracle_should_calc_lambda(idx, level, k)
            assert bool(cl) == RL.ndlqr_ShouldCalcLambda(tree, idx, k)
            for up in range(level + 1, K):
                OL.oracle_update_schur(o.h, 0, idx, k, level, up, cl)
                RL.ndlqr_UpdateShurFactor(rfact, rfact, idx, k, level, up, bool(cl))
        assert np.array_equal(o.fact(), r.fact())
    for k in range(N - 1):
        assert OL.oracle_index_level(k) == RL.ndlqr_GetIndexLevel(tree, k)


@pytest.mark.parametrize("threads", [1, 3, 8])
def test_full_solve_bit_identical(oracle, ref, ndlqr, threads):
    prob = synth(ndlqr, 12, 4, 64, 5)
    z, fact, _, _ = oracle.solve(prob, threads, want_fact=True)
    rs = ref.solver(prob)
    rs.solve(threads)
    assert np.array_equal(z, rs.soln()) and np.array_equal(fact, rs.fact())
    res, bnorm = oracle.kkt_residual(prob, z[: prob.nvars])
    assert res <= 1e-9 * max(1.0, bnorm)


@pytest.mark.parametrize("fname", ["lqr_prob.json", "lqr_prob_256.json"])
def test_reference_riccati_on_the_fixtures(fname):
    """the reference's serial Riccati solver (src/riccati_solve.c:26-150), compiled into
    oracle/_ref, is the second CPU column of bench.py. On the two JSON fixtures it reproduces the stored
    solution (the reference's own check: test/riccati_solver_test.c:332-370, test/sample_problem_test.c:150-151)."""
    from support import GOLDEN, load_json_problem
    import os
    prob, soln = load_json_problem(os.path.join(GOLDEN, fname))
    x, ms = Reference().riccati(prob, reps=3)
    assert ms > 0
    assert np.linalg.norm(x - soln) < 1e-6
    z = Reference().solver(prob)
    z.solve(1)
    assert np.linalg.norm(x - z.soln()[: prob.nvars]) < 1e-6  # "same answer" as rsLQR (sample_problem_test.c:151)

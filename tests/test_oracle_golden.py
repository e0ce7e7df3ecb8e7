"""Pins the CPU oracle (oracle/ndlqr_oracle.c) to the reference's own golden vectors and to
fixtures generated from the real reference (tests/golden/make_golden.py). Runs without a GPU and
without /root/reference.

Golden sources (all under /root/reference, copied as data into tests/golden/):
  lqr_prob.json["soln"], lqr_prob_256.json["soln"]       test/sample_problem_test.c:150-151
  sample_problem.json  b, E01*, E11*, E02*, E12*, soln   test/nested_dissection_test.c:103-105,
                                                         197-199, 227-229, 277
  literals of test/nested_dissection_test.c              :49-52, 72-75 (rhs), :125-133 (S-bar),
                                                         :155-180 (f before/after the solve)
"""
import os

import numpy as np
import pytest

from support import GOLDEN, Problem, load_json_matrix, load_json_problem, split_block

SAMPLE = os.path.join(GOLDEN, "sample_problem.json")
TOL = 1e-6  # the reference's own absolute tolerance


@pytest.fixture()
def test_solver(oracle):
    prob, _ = load_json_problem(os.path.join(GOLDEN, "lqr_prob.json"))
    return prob, oracle.solver(prob)


def leaves(oracle, s, prob):
    for k in range(prob.N):
        oracle.L.oracle_solve_leaf(s.h, k)


def test_final_solution_matches_json(oracle):
    for fname, tol in (("lqr_prob.json", 1e-10), ("lqr_prob_256.json", 1e-6)):
        prob, soln = load_json_problem(os.path.join(GOLDEN, fname))
        z, _, _, fails = oracle.solve(prob, 1)
        assert fails == 0
        assert np.linalg.norm(z[: prob.nvars] - soln) < tol
        assert np.linalg.norm(z[: prob.nvars] - soln) / np.linalg.norm(soln) < 1e-13
    sample_soln = load_json_matrix(SAMPLE, "soln").ravel()
    prob, _ = load_json_problem(os.path.join(GOLDEN, "lqr_prob.json"))
    z, _, _, _ = oracle.solve(prob, 1)
    assert np.linalg.norm(z[: prob.nvars] - sample_soln) < TOL


def test_threads_do_not_change_the_answer(oracle):
    prob, _ = load_json_problem(os.path.join(GOLDEN, "lqr_prob_256.json"))
    z1, f1, _, _ = oracle.solve(prob, 1, want_fact=True)
    z4, f4, _, _ = oracle.solve(prob, 4, want_fact=True)
    assert np.array_equal(z1, z4) and np.array_equal(f1, f4)


def test_leaves_literals(oracle, test_solver):
    prob, s = test_solver
    leaves(oracle, s, prob)
    n, m = prob.n, prob.m
    A0 = prob.A[0].reshape(n, n).T  # (rows, cols)
    B0 = prob.B[0].reshape(m, n).T
    Fy, Fx, Fu = s.fact_block(0, 0)
    assert np.linalg.norm(Fy - (-A0.T)) < TOL
    assert np.linalg.norm(Fx) < TOL
    assert np.linalg.norm(Fu - (B0 / prob.R[0][0]).T) < TOL
    z = s.soln()
    z0 = [-1.0, -2.2, 1.6, -1.6, 4.2, -1.0, 1.0, -1.0, 2.0, -2.0, 3.0, -3.0, 100.0, -0.0, -100.0]
    z1 = [-1.5, -1.5, -1.5, -1.5, -1.5, -1.5, 4.0, 2.4, 0.8, -0.8, -2.4, -4.0, 200.0, -0.0, -200.0]
    assert np.linalg.norm(z[:15] - z0) < TOL
    assert np.linalg.norm(z[15:30] - z1) < TOL
    _, Fx1, Fu1 = s.fact_block(1, 1)
    assert np.linalg.norm(Fx1 - A0.T) < TOL
    assert np.linalg.norm(Fu1 - (B0 / prob.R[0][0]).T) < TOL
    _, Fx10, _ = s.fact_block(1, 0)
    assert np.linalg.norm(Fx10 - np.diag(-1.0 / prob.Q[1])) < TOL
    # rhs after the leaf phase == sample_problem.json["b"]
    b = load_json_matrix(SAMPLE, "b").ravel()
    assert np.linalg.norm(z[: b.size] - b) < TOL


def test_inner_product_and_schur_literals(oracle, test_solver):
    prob, s = test_solver
    L = oracle.L
    leaves(oracle, s, prob)
    L.oracle_inner_product(s.h, 0, 0, 0, 0)
    S, _, _ = s.fact_block(1, 0)
    Sans = np.array([[1.0025, 0, 0, 0.05, 0, 0], [0, 1.0025, 0, 0, 0.05, 0], [0, 0, 1.0025, 0, 0, 0.05],
                     [0.05, 0, 0, 2.0, 0, 0], [0, 0.05, 0, 0, 2.0, 0], [0, 0, 0.05, 0, 0, 2.0]])
    assert np.linalg.norm(S - Sans) < TOL
    L.oracle_factor_separator(s.h, 0, 0)
    L.oracle_inner_product(s.h, 0, 0, 0, 1)
    f, _, _ = s.fact_block(1, 1)
    fans = -np.eye(6)
    fans[3:, :3] = 0
    fans[:3, 3:] = -0.1 * np.eye(3)  # literal is written column-wise in the reference test
    assert np.linalg.norm(f - fans.T) < TOL
    L.oracle_solve_chol_factor(s.h, 0, 0, 1)
    f, _, _ = s.fact_block(1, 1)
    f2 = np.zeros((6, 6))
    for i in range(3):
        f2[i, i] = -0.996255; f2[i + 3, i] = -0.0250936   # column i of the literal block
        f2[i, i + 3] = 0.0249688; f2[i + 3, i + 3] = -0.500624
    assert np.linalg.norm(f - f2) < 1e-5
    L.oracle_compute_schur_compliment(s.h, 0, 0, 1)
    for i in range(2):
        lam, st, inp = s.fact_block(i, 1)
        for blk, tag in ((lam, "y"), (st, "x"), (inp, "u")):
            assert np.linalg.norm(blk - load_json_matrix(SAMPLE, "E%d1%s" % (i, tag))) < TOL
    L.oracle_inner_product(s.h, 0, 0, 0, 2)
    L.oracle_solve_chol_factor(s.h, 0, 0, 2)
    L.oracle_compute_schur_compliment(s.h, 0, 0, 2)
    for i in range(2):
        lam, st, inp = s.fact_block(i, 2)
        for blk, tag in ((lam, "y"), (st, "x"), (inp, "u")):
            assert np.linalg.norm(blk - load_json_matrix(SAMPLE, "E%d2%s" % (i, tag))) < TOL


def test_final_top_level_blocks_match_sample(oracle):
    """F{k}2{y,x,u}: final top-level factor column (assertions are commented out in the
    reference test, nested_dissection_test.c:265-267, but the data is valid)."""
    prob, _ = load_json_problem(os.path.join(GOLDEN, "lqr_prob.json"))
    _, fact, _, _ = oracle.solve(prob, 1, want_fact=True)
    for k in range(prob.N):
        lam, st, inp = split_block(fact, prob, k, prob.K - 1)
        for blk, tag in ((st, "x"), (inp, "u")):
            gold = load_json_matrix(SAMPLE, "F%d2%s" % (k, tag))
            assert np.linalg.norm(blk - gold) < TOL, (k, tag)


def test_synthetic_fixtures_from_reference(oracle, ndlqr):
    """Bit-exact against vectors produced by the real reference (make_golden.py)."""
    gold = np.load(os.path.join(GOLDEN, "synthetic_ref.npz"))
    for (n, m, N, seed) in gold["cases"]:
        n, m, N, seed = int(n), int(m), int(N), int(seed)
        g = ndlqr.generate_synthetic(n, m, N, seed)
        prob = Problem(n, m, N, g["A"], g["B"], g["Q"], g["R"], g["q"], g["r"], g["d"], g["x0"])
        z, fact, _, fails = oracle.solve(prob, 1, want_fact=True)
        key = "n%d_m%d_N%d_s%d" % (n, m, N, seed)
        assert fails == 0
        assert np.array_equal(z[: prob.nvars], gold[key + "_soln"]), key
        K = prob.K
        for (k, lvl) in [(0, K - 1), (N // 2, K - 1), (N - 1, K - 1), (1, 0)]:
            lam, st, inp = split_block(fact, prob, k, lvl)
            got = np.concatenate([lam.ravel(), st.ravel(), inp.ravel()])
            assert np.array_equal(got, gold[key + "_F%d_%d" % (k, lvl)]), (key, k, lvl)
        assert np.allclose([fact.sum(), (fact * fact).sum()], gold[key + "_factsum"], rtol=1e-13)
        res, bnorm = oracle.kkt_residual(prob, z[: prob.nvars])
        assert res <= 1e-9 * max(1.0, bnorm)
        assert np.allclose([res, bnorm], gold[key + "_kkt"], rtol=1e-6, atol=1e-18)

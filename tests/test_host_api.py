"""Host-side behaviour of the drop-in C API (no GPU needed): containers, tree closed forms, NdData
layout, JSON readers, KKT assembly -- the reference's unit tests replayed through ctypes
(test/binarytree_test.c, test/nddata_test.c, test/utils_test.c, test/lqrdata_test.c,
test/solver_test.c) -- plus: every symbol declared in include/*.h is exported, and the solver
fails loudly (no CPU fallback) when no HIP device is present."""
import ctypes as C
import os

import numpy as np
import pytest

from support import GOLDEN, load_json_problem

LQRPROB = os.path.join(GOLDEN, "lqr_prob.json").encode()
LQRDATA = os.path.join(GOLDEN, "lqr_data.json").encode()
SAMPLE = os.path.join(GOLDEN, "sample_problem.json").encode()


@pytest.fixture(scope="module")
def L(ndlqr):
    return ndlqr.lib()


def test_every_declared_symbol_is_exported(ndlqr, L):
    names = ndlqr.exported_symbols()
    assert len(names) > 100
    missing = [s for s in names if not hasattr(L, s)]
    assert not missing, missing
    assert b"gfx950" in L.ndlqr_Version()


def test_build_tree(ndlqr, L):  # test/binarytree_test.c:4-21
    tree = L.ndlqr_BuildTree(8)
    assert tree.num_elements == 8 and tree.depth == 3
    root = tree.root.contents
    assert (root.idx, root.level) == (3, 2)
    assert (root.left_inds.start, root.left_inds.stop) == (0, 3)
    assert (root.right_inds.start, root.right_inds.stop) == (4, 7)
    lc = root.left_child.contents
    assert (lc.idx, lc.level) == (1, 1)
    assert (lc.left_child.contents.idx, lc.left_child.contents.level) == (0, 0)
    assert root.right_child.contents.idx == 5
    assert root.right_child.contents.right_child.contents.idx == 6
    assert lc.parent.contents.idx == 3 and lc.left_child.contents.parent.contents.idx == 1
    for k, lvl in enumerate([0, 1, 0, 2, 0, 1, 0]):  # test/binarytree_test.c:23-34
        assert L.ndlqr_GetIndexLevel(C.byref(tree), k) == lvl
    for (idx, lvl, want) in [(5, 0, 4), (3, 0, 2), (2, 2, 3), (7, 2, 3), (7, 0, 6)]:  # :36-60
        assert L.ndlqr_GetIndexAtLevel(C.byref(tree), idx, lvl) == want
    assert L.ndlqr_GetIndexFromLeaf(C.byref(tree), 1, 1) == 5
    L.ndlqr_FreeTree(C.byref(tree))


def test_tree_matches_closed_forms_for_large_horizon(L):
    tree = L.ndlqr_BuildTree(64)
    for k in range(63):
        nd = tree.node_list[k]
        lvl = (~k & (k + 1)).bit_length() - 1
        assert nd.level == lvl
        assert (nd.left_inds.start, nd.left_inds.stop) == (k - (1 << lvl) + 1, k)
        assert (nd.right_inds.start, nd.right_inds.stop) == (k + 1, k + (1 << lvl))
        assert L.ndlqr_ShouldCalcLambda(C.byref(tree), k, 0)
        assert not L.ndlqr_ShouldCalcLambda(C.byref(tree), k, k + 1)
    L.ndlqr_FreeTree(C.byref(tree))


def test_nddata_layout(ndlqr, L):  # test/nddata_test.c:11-126
    assert not L.ndlqr_NewNdData(0, 3, 8, 6)
    assert not L.ndlqr_NewNdData(6, 0, 8, 6)
    assert not L.ndlqr_NewNdData(6, 3, 1, 6)
    assert not L.ndlqr_NewNdData(6, 3, 7, 6)  # not a power of two
    nd = L.ndlqr_NewNdData(6, 3, 8, 6)
    d = nd.contents
    assert (d.nstates, d.ninputs, d.nsegments, d.depth, d.width) == (6, 3, 7, 3, 6)
    fsize = (2 * 6 + 3) * 6
    base = C.addressof(d.data.contents)
    f = C.POINTER(ndlqr.NdFactor)()
    assert L.ndlqr_GetNdFactor(nd, 1, 0, C.byref(f)) == 0
    assert C.addressof(f.contents.lambda_.data.contents) == base + 8 * fsize
    assert C.addressof(f.contents.state.data.contents) == base + 8 * (fsize + 36)
    assert C.addressof(f.contents.input.data.contents) == base + 8 * (fsize + 72)
    assert (f.contents.input.rows, f.contents.input.cols) == (3, 6)
    assert L.ndlqr_GetNdFactor(nd, 0, 1, C.byref(f)) == 0
    assert C.addressof(f.contents.lambda_.data.contents) == base + 8 * fsize * 8  # level stride N
    assert L.ndlqr_GetNdFactor(nd, 8, 0, C.byref(f)) == -1
    assert L.ndlqr_GetNdFactor(nd, 0, 3, C.byref(f)) == -1
    d.numpy()[:] = 1.5
    L.ndlqr_ResetNdData(nd)
    assert not d.numpy().any()
    L.ndlqr_FreeNdData(nd)
    rhs = L.ndlqr_NewNdData(6, 3, 8, 1)  # width 1 -> one column (src/nddata.c:23-29)
    assert rhs.contents.depth == 1 and rhs.contents.numpy().size == 8 * 15
    L.ndlqr_FreeNdData(rhs)


def test_read_file_and_matrix(L):  # test/utils_test.c:17-51
    data, n = C.c_char_p(), C.c_int()
    assert L.ReadFile(LQRDATA, C.byref(data), C.byref(n)) == 0
    assert n.value == 473 and data.value[:3] == b'{"i' and len(data.value) == 473
    assert L.ReadFile(b"/nonexistent/file.json", C.byref(data), C.byref(n)) == -1
    mat = L.ReadMatrixJSONFile(SAMPLE, b"test")
    assert [mat.data[i] for i in range(12)] == [float(i + 1) for i in range(12)]
    L.FreeMatrix(C.byref(mat))
    none = L.ReadMatrixJSONFile(SAMPLE, b"no_such_field")
    assert none.rows == 0 and not none.data


def check_lqrdata(ld):  # test/lqrdata_test.c:15-43
    assert (ld.nstates, ld.ninputs) == (6, 3)
    assert [ld.Q[i] for i in range(6)] == [1.0] * 6
    assert [ld.R[i] for i in range(3)] == [0.01] * 3
    A = np.array([ld.A[i] for i in range(36)]).reshape(6, 6).T
    B = np.array([ld.B[i] for i in range(18)]).reshape(3, 6).T
    assert np.array_equal(np.diag(A), np.ones(6))
    for i in range(3):
        assert abs(A[i, i + 3] - 0.1) < 1e-8 and abs(B[i + 3, i] - 0.1) < 1e-8


def test_read_lqrdata_and_problem(L):  # test/lqrdata_test.c:45-100
    ld = L.ndlqr_ReadLQRDataJSONFile(LQRDATA)
    check_lqrdata(ld.contents)
    L.ndlqr_FreeLQRData(ld)
    prob = L.ndlqr_ReadLQRProblemJSONFile(LQRPROB)
    p = prob.contents
    assert p.nhorizon == 8
    assert [p.x0[i] for i in range(6)] == [1, -1, 2, -2, 3, -3]
    for k in range(7):
        check_lqrdata(p.lqrdata[k].contents)
    py, _ = load_json_problem(LQRPROB.decode())
    for k in range(8):
        ld = p.lqrdata[k].contents
        assert np.array_equal([ld.A[i] for i in range(36)], py.A[k])
        assert np.array_equal([ld.B[i] for i in range(18)], py.B[k])
        assert np.array_equal([ld.d[i] for i in range(6)], py.d[k])
        assert np.array_equal([ld.q[i] for i in range(6)], py.q[k])
    L.ndlqr_FreeLQRProblem(prob)
    assert not L.ndlqr_ReadLQRProblemJSONFile(b"/nonexistent.json")


def test_initialize_with_lqr_problem(ndlqr, L, oracle):  # test/solver_test.c:20-151
    prob = L.ndlqr_ReadLQRProblemJSONFile(LQRPROB)
    solver = L.ndlqr_NewNdLqrSolver(6, 3, 8)
    s = solver.contents
    assert s.nvars == 117 and s.depth == 3
    assert L.ndlqr_GetIndexLevel(C.byref(s.tree), 0) == 0 and L.ndlqr_GetIndexLevel(C.byref(s.tree), 3) == 2
    assert L.ndlqr_InitializeWithLQRProblem(prob, solver) == 0
    py, _ = load_json_problem(LQRPROB.decode())
    o = oracle.solver(py)  # oracle_initialize is pinned to the reference in test_oracle_vs_reference
    assert np.array_equal(s.data.contents.numpy(), o.data())
    assert np.array_equal(s.soln.contents.numpy(), o.soln())
    diag = np.ctypeslib.as_array(s.diagonals[0].data, (8 * (36 + 9),))
    assert np.array_equal(diag, np.ctypeslib.as_array(oracle.L.oracle_diag(o.h), (8 * 45,)))
    # spot checks written out like the reference test
    f = C.POINTER(ndlqr.NdFactor)()
    L.ndlqr_GetNdFactor(s.data, 0, 0, C.byref(f))
    A0 = py.A[0].reshape(6, 6).T
    assert np.array_equal(f.contents.state.numpy(), A0.T)
    L.ndlqr_GetNdFactor(s.data, 1, 0, C.byref(f))
    assert np.array_equal(f.contents.state.numpy(), -np.eye(6))
    L.ndlqr_GetNdFactor(s.data, 7, 0, C.byref(f))
    assert np.array_equal(f.contents.state.numpy(), -np.eye(6))
    L.ndlqr_GetNdFactor(s.soln, 1, 0, C.byref(f))
    assert np.array_equal(f.contents.lambda_.numpy().ravel(), -py.d[0])
    # dimension / horizon mismatches are rejected (src/solver.c:125,142-143)
    other = L.ndlqr_NewNdLqrSolver(6, 3, 16)
    assert L.ndlqr_InitializeWithLQRProblem(prob, other) == -1
    L.ndlqr_FreeNdLqrSolver(other)
    other = L.ndlqr_NewNdLqrSolver(5, 3, 8)
    assert L.ndlqr_InitializeWithLQRProblem(prob, other) == -1
    L.ndlqr_FreeNdLqrSolver(other)
    assert not L.ndlqr_NewNdLqrSolver(6, 3, 12)
    # reset clears mirrors, thread setters behave like the reference
    L.ndlqr_ResetSolver(solver)
    assert not s.data.contents.numpy().any() and not s.soln.contents.numpy().any()
    assert L.ndlqr_SetNumThreads(solver, 4) == 0 and L.ndlqr_GetNumThreads(solver) == 4
    assert L.ndlqr_SetNumThreads(None, 4) == -1
    L.ndlqr_FreeLQRProblem(prob)
    L.ndlqr_FreeNdLqrSolver(solver)


def test_synthetic_generator_is_reproducible(ndlqr):
    a = ndlqr.generate_synthetic(12, 4, 16, 42)
    b = ndlqr.generate_synthetic(12, 4, 16, 42)
    c = ndlqr.generate_synthetic(12, 4, 16, 43)
    assert all(np.array_equal(a[k], b[k]) for k in a)
    assert not np.array_equal(a["A"], c["A"])
    assert (a["Q"] >= 0.5).all() and (a["Q"] <= 2.0).all()
    assert (a["R"] >= 0.01).all() and (a["R"] <= 0.1).all()
    A0 = a["A"][0].reshape(12, 12).T
    assert np.allclose(A0 + A0.T, 2 * 0.99 * np.eye(12))  # (1 - h g) I + skew part


def test_solver_fails_loudly_without_a_device(ndlqr, L):
    if ndlqr.device_count() > 0:
        pytest.skip("a HIP device is present")
    prob = L.ndlqr_ReadLQRProblemJSONFile(LQRPROB)
    solver = L.ndlqr_NewNdLqrSolver(6, 3, 8)
    assert L.ndlqr_InitializeWithLQRProblem(prob, solver) == 0
    before = solver.contents.soln.contents.numpy().copy()
    assert L.ndlqr_Solve(solver) == -2  # NDLQR_ERR_NO_DEVICE, nothing computed on the CPU
    assert np.array_equal(solver.contents.soln.contents.numpy(), before)
    assert not L.ndlqr_NewBatchSolver(6, 3, 8, 4, -1)
    with pytest.raises(RuntimeError):
        ndlqr.BatchSolver(6, 3, 8, 4)
    x = np.eye(3)
    p = x.ctypes.data_as(C.POINTER(C.c_double))
    assert L.ndlqr_hip_potrf_lower(3, p, 3) == -2  # dense helpers have no CPU fallback either
    L.ndlqr_FreeLQRProblem(prob)
    L.ndlqr_FreeNdLqrSolver(solver)


def test_batch_limit_is_reported(ndlqr, L):
    """The batch index rides on gridDim.y: more than 65 535 problems per solver are refused up front
    with a clear message (not at the first launch, and not as 'no device')."""
    assert not L.ndlqr_NewBatchSolver(4, 1, 8, 70000, -1)
    assert b"65535" in L.ndlqr_hip_last_error()


def _mutated_fixture(tmp_path, name, mutate):
    import json
    with open(os.path.join(GOLDEN, "lqr_prob.json")) as fh:
        j = json.load(fh)
    text = mutate(j)
    p = tmp_path / name
    p.write_text(text if isinstance(text, str) else json.dumps(j))
    return str(p).encode()


def test_malformed_problem_files_return_null(L, tmp_path):
    """A problem file the reference would read out of bounds with (src/json_utils.c:186-259 trusts `nhorizon`,
    `nstates` and `index`; src/lqr_data.c:24-49 allocates n * n in int arithmetic) comes back as NULL here --
    truncated text, wrong dimensions, nstates = 60000, a knot index out of range, a knot missing, a horizon longer
    than the file -- never as a crash or as a problem with uninitialised knots."""
    import json
    good = L.ndlqr_ReadLQRProblemJSONFile(LQRPROB)
    assert good
    L.ndlqr_FreeLQRProblem(good)

    def truncated(j):
        return json.dumps(j)[: len(json.dumps(j)) // 2]

    def huge_states(j):
        j["lqrdata"][0]["nstates"] = 60000

    def wrong_dims(j):
        j["lqrdata"][3]["nstates"] = 5

    def short_A(j):
        j["lqrdata"][2]["A"] = j["lqrdata"][2]["A"][:-1]

    def index_out_of_range(j):
        j["lqrdata"][4]["index"] = 1000

    def negative_index(j):
        j["lqrdata"][4]["index"] = -3

    def duplicate_index(j):
        j["lqrdata"][4]["index"] = j["lqrdata"][5]["index"]

    def long_horizon(j):
        j["nhorizon"] = 1 << 20

    def huge_horizon(j):
        j["nhorizon"] = 1e300

    def zero_inputs(j):
        for kd in j["lqrdata"]:
            kd["ninputs"] = 0

    def non_number(j):
        j["lqrdata"][1]["Q"][2] = "x"

    def short_x0(j):
        j["x0"] = j["x0"][:-1]

    def no_lqrdata(j):
        del j["lqrdata"]

    for mutate in (truncated, huge_states, wrong_dims, short_A, index_out_of_range, negative_index, duplicate_index,
                   long_horizon, huge_horizon, zero_inputs, non_number, short_x0, no_lqrdata):
        path = _mutated_fixture(tmp_path, mutate.__name__ + ".json", mutate)
        prob = L.ndlqr_ReadLQRProblemJSONFile(path)
        assert not prob, mutate.__name__
    assert not L.ndlqr_ReadLQRProblemJSONFile(b"/nonexistent/file.json")
    (tmp_path / "empty.json").write_text("")
    assert not L.ndlqr_ReadLQRProblemJSONFile(str(tmp_path / "empty.json").encode())
    (tmp_path / "deep.json").write_text("[" * 5000)
    assert not L.ndlqr_ReadLQRProblemJSONFile(str(tmp_path / "deep.json").encode())


def test_container_constructors_reject_bad_sizes(L):
    """ndlqr_NewLQRData / ndlqr_NewLQRProblem (src/lqr_data.c:24-49, src/lqr_problem.c:7-30): non-positive or absurd
    block sizes give NULL (the slab size is computed in size_t; 60 000 states would wrap an int)."""
    for n, m in ((0, 3), (3, 0), (-1, 2), (60000, 60000), (1 << 30, 1)):
        assert not L.ndlqr_NewLQRData(n, m), (n, m)
    assert not L.ndlqr_NewLQRProblem(6, 3, 0)
    assert not L.ndlqr_NewLQRProblem(0, 3, 8)
    assert not L.ndlqr_NewLQRProblem(1 << 30, 3, 8)
    big = L.ndlqr_NewLQRData(2000, 10)  # 32 MB: fine
    assert big
    assert big.contents.A[2000 * 2000 - 1] == 0.0  # zero-initialised slab, last entry of A addressable
    L.ndlqr_FreeLQRData(big)

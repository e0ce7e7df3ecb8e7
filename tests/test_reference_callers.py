"""The reference's own test programs, compiled unchanged against the drop-in.

`oracle/Makefile` target `reftests` compiles test/{matrix,utils,binarytree,lqrdata,nddata,solver,
linalg,linalg_custom,nested_dissection,riccati_solver,sample_problem,parallel}_test.c of the reference -- every test
program its own CMake builds with the default backend -- from where they lie (never copied) against
`include/` and links them with `rslqr_amd/librslqr_amd.so`; the binaries land in
`oracle/_ref/tests/` (git-ignored, they travel to the GPU box like `oracle/_ref/libref.so`).
Their fixture macros (test/CMakeLists.txt:42-46 of the reference) point at the data copies in
`tests/golden/`, so they run from the repo root.

* host-only programs (containers, tree, JSON readers) run on the CPU here;
* `solver_test`, `linalg_test`, `linalg_custom_test` (the `clap_*` names), `nested_dissection_test`,
  `riccati_solver_test`, `sample_problem_test` and `parallel_test` (stage functions and MatrixMultiply from an
  OpenMP team) run on the GPU: they call the stage functions, the dense helpers,
  `ndlqr_Solve` and the Riccati baseline, and assert the reference's literal golden
  values (nested_dissection_test.c:44-105,133,166-229,277,299,307; linalg_test.c:19,55).
"""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "oracle", "_ref", "tests")
HOST_ONLY = ["matrix", "utils", "binarytree", "lqrdata", "nddata"]
DEVICE = ["solver", "linalg", "linalg_custom", "nested_dissection", "riccati_solver"]


def _build_if_possible():
    if os.path.isdir("/root/reference/test"):
        import rslqr_amd.build as build

        build.build()
        subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "reftests"], check=True,
                       capture_output=True)


def _run(name):
    exe = os.path.join(BIN, name + "_test")
    if not os.path.exists(exe):
        pytest.skip("oracle/_ref/tests not built (needs /root/reference at build time)")
    out = subprocess.run([exe], cwd=ROOT, capture_output=True, text=True, timeout=300)
    text = out.stdout + out.stderr
    assert "TEST FAILED" not in text, text[-3000:]
    assert "ALL TESTS PASSED!" in text, text[-3000:]
    assert out.returncode == 0, text[-3000:]
    return text


@pytest.fixture(scope="module")
def built():
    _build_if_possible()


@pytest.mark.parametrize("name", HOST_ONLY)
def test_reference_host_program(built, name):
    _run(name)


def test_reference_device_programs_link(built):
    """On the CPU box the device programs must at least have linked against the drop-in."""
    if not os.path.isdir("/root/reference/test"):
        pytest.skip("reference absent")
    for name in DEVICE + ["sample_problem", "parallel"]:
        assert os.access(os.path.join(BIN, name + "_test"), os.X_OK), name
    for name in ("importexample", "installexample"):  # the reference's example callers
        assert os.access(os.path.join(BIN, name + "_example"), os.X_OK), name


@pytest.mark.gpu
@pytest.mark.parametrize("name", HOST_ONLY + DEVICE)
def test_reference_program_on_gpu(name):
    text = _run(name)
    if name == "nested_dissection":
        # the program prints its own final-solution error against lqr_prob.json's soln
        assert "Accuracy of final solution" in text


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["importexample", "installexample"])
def test_reference_example_program(name):
    """examples/*/main.c of the reference -- read a problem, New / Initialize / Solve / PrintSolveSummary /
    Free -- compiled unchanged against include/ and run against the drop-in."""
    exe = os.path.join(BIN, name + "_example")
    if not os.path.exists(exe):
        pytest.skip("oracle/_ref/tests not built (needs /root/reference at build time)")
    out = subprocess.run([exe], cwd=ROOT, capture_output=True, text=True, timeout=300)
    text = out.stdout + out.stderr
    assert out.returncode == 0, text[-3000:]
    assert "rsLQR package" in text
    assert "rsLQR Solve Summary" in text and "Device solve time" in text, text[-2000:]


@pytest.mark.gpu
@pytest.mark.parametrize("length", ["8", "256"])
def test_reference_sample_problem_program(length):
    """test/sample_problem_test.c (the reference's end-to-end harness: rsLQR against the stored
    solution and against its Riccati baseline, N = 8 and N = 256): it only prints, so the two
    verdict lines are checked here."""
    exe = os.path.join(BIN, "sample_problem_test")
    if not os.path.exists(exe):
        pytest.skip("oracle/_ref/tests not built (needs /root/reference at build time)")
    out = subprocess.run([exe, length], cwd=ROOT, capture_output=True, text=True, timeout=600)
    text = out.stdout + out.stderr
    assert out.returncode == 0, text[-3000:]
    assert "Using a trajectory length of %s" % length in text
    assert "Got the right answer? 1" in text, text[-3000:]
    assert "Got the same answer? 1" in text, text[-3000:]



@pytest.mark.gpu
def test_reference_parallel_program():
    """test/parallel_test.c:30-239 -- the reference's thread-scaling harness: ndlqr_SolveLeaf, the inner products
    and MatrixMultiply called from an OpenMP team (here two threads, concurrently, each call on its own pooled
    device scratch), then serial-vs-parallel ndlqr_Solve with ndlqr_CompareProfile. It only prints (returns 0,
    test/parallel_test.c:248-256): the run must complete with all four of its tests passing."""
    exe = os.path.join(BIN, "parallel_test")
    if not os.path.exists(exe):
        pytest.skip("oracle/_ref/tests not built (needs /root/reference at build time)")
    env = dict(os.environ, OMP_NUM_THREADS="2")
    out = subprocess.run([exe], cwd=ROOT, capture_output=True, text=True, timeout=900, env=env)
    text = out.stdout + out.stderr
    assert out.returncode == 0, text[-3000:]
    assert "TEST FAILED" not in text, text[-3000:]
    assert "ALL TESTS PASSED!" in text, text[-3000:]
    assert "Actual number of threads = " in text and "rsLQR Solve Summary" in text, text[-3000:]

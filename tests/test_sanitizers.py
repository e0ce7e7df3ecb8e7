"""Sanitizer run of the host side: `make -C oracle asan-check` builds the seven plain-C files of the
product library and the oracle with -fsanitize=address,undefined (CPU build only: the HIP translation units are linked
in as hipcc built them and never run here) and runs, against those libraries, the reference's five host-only test
programs (compiled unchanged, where /root/reference exists), tests/test_host_api.py -- with the malformed-JSON cases --
and tests/test_oracle_golden.py."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_host_side_under_asan_and_ubsan():
    if os.environ.get("NDLQR_LIBRARY"):
        pytest.skip("already inside the sanitizer run")
    gcc = shutil.which("gcc")
    if not gcc or not os.path.exists(subprocess.run([gcc, "-print-file-name=libasan.so"], capture_output=True,
                                                    text=True).stdout.strip()):
        pytest.skip("gcc without libasan")
    import rslqr_amd.build as build
    build.build()  # the HIP objects the sanitised library links in
    env = dict(os.environ)
    env.pop("LD_PRELOAD", None)
    proc = subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "asan-check"], capture_output=True,
                          text=True, env=env, timeout=900)
    tail = (proc.stdout + proc.stderr)[-4000:]
    assert proc.returncode == 0, tail
    assert "passed" in proc.stdout and "failed" not in proc.stdout, tail
    for t in ("matrix", "utils", "binarytree", "lqrdata", "nddata"):
        log = os.path.join(ROOT, "oracle", "_asan", t + ".log")
        if os.path.isdir("/root/reference/test"):
            text = open(log).read()
            assert "ALL TESTS PASSED!" in text and "runtime error" not in text and "AddressSanitizer" not in text, text[-2000:]

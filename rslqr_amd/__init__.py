"""rslqr_amd -- MI355X-native nested-dissection LQR solver behind the rsLQR `ndlqr_*` C API.

The product is the shared library ``librslqr_amd.so`` (plain-C host code + HIP kernels for
gfx950, see ``csrc/`` and ``include/ndlqr.h``). This Python package is only the host-side mirror
used by tests and the benchmark: ctypes bindings with the same names and argument meaning as the
C API, plus a numpy-friendly ``BatchSolver``.
"""
from .api import (  # noqa: F401
    BatchSolver,
    DeviceArray,
    FLAG_GENERIC,
    FLAG_KEEP_FACT,
    FLAG_KEEP_RECORDS,
    FLAG_PROFILE,
    FLAG_STRICT_FP,
    LQRData,
    LQRProblem,
    Matrix,
    NdData,
    NdFactor,
    NdLqrSolver,
    SOLN_INPUT,
    SOLN_ONLY,
    SOLN_LAMBDA,
    SOLN_STATE,
    device_count,
    exported_symbols,
    generate_synthetic,
    lib,
    library_path,
    pinned_empty,
)

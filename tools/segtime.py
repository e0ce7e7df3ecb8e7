"""Developer tool: where do the cycles of bottom_small / level_small go?

Builds (here, on the CPU box) an instrumented copy of the library with -DNDLQR_SEGTIME -- marks
of s_memtime between the phases of the kernels, summed over lane 0 of every wavefront -- and, on
the GPU box, runs the default workload through it:

    python tools/segtime.py --build          # here
    gpurun -- python tools/segtime.py        # there

Prints average cycles per wavefront per segment. (Elapsed cycles, not issue cycles: the SIMDs issue
oldest-wavefront-first, so the first segments of a wavefront -- staging, leaf tiles -- look long
because a young wavefront only gets the slots its elders leave.) The instrumented library is never shipped or
loaded by the package itself (NDLQR_LIBRARY points the ctypes mirror at it for this run only).
"""
import ctypes as C
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
SEGLIB = os.path.join(ROOT, "rslqr_amd", "librslqr_amd_seg.so")

NAMES = {0: "level: stage operands", 1: "level core: P1", 2: "level core: P2 (Cholesky)",
         3: "level core: P3 (substitutions)", 6: "level: (re-arm)", 4: "level: record stores",
         5: "level: boundary Schur rows",
         9: "reduced: assemble (loads, S-bar tile, panel)", 10: "reduced: Cholesky", 11: "reduced: substitutions + Gram pushes",
         13: "reduced: (re-arm)", 12: "reduced: record stores + drain",
         17: "bottom core: P1", 18: "bottom core: P2", 19: "bottom core: P3",
         20: "bottom: stage AB", 21: "bottom: leaf", 22: "bottom: publish + barrier",
         23: "bottom: separator (owner) / skip", 24: "bottom: barrier after separator",
         30: "mc core: S-bar rows, Cholesky + inverse", 31: "mc core: W to operands, S^-1 = W'W",
         32: "mc core: X = S^-1 R", 33: "mc core: hook (Schur blocks R'X, pushes)", 34: "mc: (re-arm)",
         35: "bottom_mc: record store", 36: "bottom_mc: last record store + drain",
         40: "generic separator: P1 products + rhs (own work)", 41: "generic separator: barrier behind P1",
         42: "generic separator: blocked Cholesky", 43: "generic separator: blocked substitutions",
         44: "generic separator: stores + drain",
         50: "reduced sep: stage [A|B], weights", 51: "reduced sep: leafS products + b~ (+ barrier)",
         52: "reduced sep: S-bar write-out (slot loads) + barrier", 53: "reduced sep: blocked Cholesky",
         54: "reduced sep: record stores, last block of y~, panel loads requested", 55: "reduced sep: column tiles Y = L^-1 R (own work)", 56: "reduced sep: barrier behind the column tiles",
         58: "reduced sep: Y_a into LDS + barrier", 59: "reduced sep: pushes",
         25: "bottom: row update + rotate", 26: "bottom: barrier end of level", 27: "bottom: hand-off"}


def build():
    import rslqr_amd.build as b
    b.build()
    objs = [os.path.join(b.OBJDIR, s + ".o") for s in b.C_SOURCES]
    o = os.path.join(b.OBJDIR, "ndlqr_hip_seg.o")
    hipcc = b._hipcc()
    # one translation unit with every instance (the counters are one __device__ array per unit)
    subprocess.run([hipcc, "--offload-arch=" + b.ARCH, "-std=c++17", "-O3", "-ffp-contract=off", "-fPIC",
                    "-DNDLQR_SEGTIME", "-DNDLQR_SINGLE_TU", "-I" + b.INCLUDE, "-I" + b.CSRC, "-c",
                    os.path.join(b.CSRC, b.HIP_MAIN), "-o", o], check=True)
    subprocess.run([hipcc, "--offload-arch=" + b.ARCH, "-shared", "-fPIC", "-o", SEGLIB] + objs + [o, "-lm"],
                   check=True)
    print("built", SEGLIB)


def main():
    if "--build" in sys.argv:
        return build()
    os.environ["NDLQR_LIBRARY"] = SEGLIB
    import rslqr_amd
    from rslqr_amd import api
    n, m, N, batch = 12, 4, 256, 1024
    if "--config5" in sys.argv:
        n, m, N, batch = 64, 16, 512, 64
    if "--horizon" in sys.argv:
        N = int(sys.argv[sys.argv.index("--horizon") + 1])
    if "--batch" in sys.argv:
        batch = int(sys.argv[sys.argv.index("--batch") + 1])
    bs = rslqr_amd.BatchSolver(n, m, N, batch)
    bs.initialize_synthetic(1)
    L = api.lib()
    L.ndlqr_hip_debug_segments.restype = C.c_int
    L.ndlqr_hip_debug_segments.argtypes = [C.POINTER(C.c_uint64), C.c_int, C.c_int]
    buf = (C.c_uint64 * 128)()
    for _ in range(3):
        bs.solve()
    L.ndlqr_hip_debug_segments(buf, 128, 1)
    steps = 5
    for _ in range(steps):
        bs.solve()
    L.ndlqr_hip_debug_segments(buf, 128, 0)
    bs.set_flags(rslqr_amd.FLAG_PROFILE)
    bs.solve()
    bs.profile_reset()
    bs.solve()
    print("instrumented kernel ms:", {k: round(v[0], 3) for k, v in bs.profile().items() if v[1]})
    for k in sorted(NAMES):
        if buf[64 + k]:
            print("%2d %-36s %10.0f cycles/wave  (%d samples)" % (k, NAMES[k], buf[k] / buf[64 + k], buf[64 + k]))


if __name__ == "__main__":
    main()

"""Builds rslqr_amd/librslqr_amd.so in-tree: plain-C host library (gcc) + HIP shim/kernels
(hipcc --offload-arch=gfx950), linked into ONE shared object that exports the ndlqr_* C API of
include/ndlqr.h and the ndlqr_hip_* shim of include/ndlqr_hip.h.

    python -m rslqr_amd.build [--force]

hipcc cross-compiles gfx950 code objects without a GPU, so this runs anywhere ROCm is installed.
"""
import os
import shutil
import subprocess
import sys

PKG = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(PKG)
CSRC = os.path.join(PKG, "csrc")
INCLUDE = os.path.join(ROOT, "include")
OBJDIR = os.path.join(PKG, "build")
LIB = os.path.join(PKG, "librslqr_amd.so")
ARCH = "gfx950"

C_SOURCES = ["containers.c", "synth.c", "json.c", "batch.c", "solver.c", "linalg.c", "stages.c"]
HIP_SOURCES = ["ndlqr_hip.hip"]
HIP_DEPS = ["kernels_common.hpp", "kernels_generic.hpp", "kernels_small.hpp", "kernels_mfma.hpp"]


def _hipcc():
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (set HIPCC)")


def _newer(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.exists(d) and os.path.getmtime(d) > t for d in deps)


def _run(cmd):
    proc = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if proc.returncode != 0:
        sys.stderr.write(" ".join(cmd) + "\n" + proc.stdout + "\n")
        raise RuntimeError("build step failed: " + cmd[0])
    if proc.stdout.strip():
        sys.stderr.write(proc.stdout)


def build(force=False, verbose=False):
    os.makedirs(OBJDIR, exist_ok=True)
    headers = [os.path.join(INCLUDE, h) for h in ("ndlqr.h", "ndlqr_hip.h")]
    me = os.path.abspath(__file__)
    objs = []
    hipcc = _hipcc()
    for src in C_SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(OBJDIR, src + ".o")
        if force or _newer(o, [s, me] + headers):
            if verbose:
                print("cc ", src)
            _run(["gcc", "-std=gnu11", "-O2", "-g0", "-fPIC", "-Wall", "-Wextra", "-I" + INCLUDE, "-c", s, "-o", o])
        objs.append(o)
    for src in HIP_SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(OBJDIR, src + ".o")
        deps = [s, me] + headers + [os.path.join(CSRC, h) for h in HIP_DEPS]
        if force or _newer(o, deps):
            if verbose:
                print("hipcc", src)
            _run([hipcc, "--offload-arch=" + ARCH, "-std=c++17", "-O3", "-ffp-contract=off", "-fPIC",
                  "-Wall", "-I" + INCLUDE, "-I" + CSRC, "-c", s, "-o", o])
        objs.append(o)
    if force or _newer(LIB, objs):
        if verbose:
            print("link", os.path.basename(LIB))
        _run([hipcc, "--offload-arch=" + ARCH, "-shared", "-fPIC", "-o", LIB] + objs + ["-lm"])
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))

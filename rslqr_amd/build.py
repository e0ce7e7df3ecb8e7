"""Builds rslqr_amd/librslqr_amd.so in-tree: plain-C host library (gcc) + HIP shim/kernels
(hipcc --offload-arch=gfx950), linked into ONE shared object that exports the ndlqr_* C API of
include/ndlqr.h and the ndlqr_hip_* shim of include/ndlqr_hip.h.

    python -m rslqr_amd.build [--force]

hipcc cross-compiles gfx950 code objects without a GPU, so this runs anywhere ROCm is installed.
"""
import concurrent.futures
import os
import re
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
HIP_MAIN = "ndlqr_hip.hip"            # context, generic / MFMA kernels, dispatch
HIP_INSTANCE = "small_instance.hip"   # compiled once per line of small_instances.def
HIP_DEPS = ["kernels_common.hpp", "kernels_leaf.hpp", "kernels_generic.hpp", "kernels_small.hpp", "kernels_bottom_reduced.hpp", "kernels_dpp.hpp", "kernels_rowbcast.hpp", "kernels_mfma.hpp", "kernels_reduced_mfma.hpp",
            "hip_context.hpp", "launch_small.hpp", "small_instances.def"]


def small_instances():
    """(nstates, ninputs) pairs listed in csrc/small_instances.def."""
    text = open(os.path.join(CSRC, "small_instances.def")).read()
    return [(int(a), int(b)) for a, b in re.findall(r"^NDLQR_SMALL_INSTANCE\((\d+),\s*(\d+)\)", text, flags=re.M)]


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
    hipcc = _hipcc()
    hip_flags = [hipcc, "--offload-arch=" + ARCH, "-std=c++17", "-O3", "-ffp-contract=off", "-fPIC", "-Wall",
                 "-Wno-pass-failed",  # (13,4) with three fused levels misses its occupancy hint: expected
                 "-I" + INCLUDE, "-I" + CSRC] + os.environ.get("NDLQR_EXTRA_HIPFLAGS", "").split()
    hip_deps = [me] + headers + [os.path.join(CSRC, h) for h in HIP_DEPS]
    jobs, objs = [], []   # (label, command) for everything out of date
    for src in C_SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(OBJDIR, src + ".o")
        if force or _newer(o, [s, me] + headers):
            jobs.append(("cc " + src, ["gcc", "-std=gnu11", "-O2", "-g0", "-fPIC", "-Wall", "-Wextra",
                                       "-I" + INCLUDE, "-c", s, "-o", o]))
        objs.append(o)
    s = os.path.join(CSRC, HIP_MAIN)
    o = os.path.join(OBJDIR, HIP_MAIN + ".o")
    if force or _newer(o, [s] + hip_deps):
        jobs.append(("hipcc " + HIP_MAIN, hip_flags + ["-c", s, "-o", o]))
    objs.append(o)
    s = os.path.join(CSRC, HIP_INSTANCE)
    for nx, nu in small_instances():
        o = os.path.join(OBJDIR, "small_%d_%d.o" % (nx, nu))
        if force or _newer(o, [s] + hip_deps):
            jobs.append(("hipcc small instance (%d,%d)" % (nx, nu),
                         hip_flags + ["-DNDLQR_INST_NX=%d" % nx, "-DNDLQR_INST_NU=%d" % nu, "-c", s, "-o", o]))
        objs.append(o)
    if jobs:
        # the translation units are independent: compile them side by side
        workers = max(1, min(len(jobs), (os.cpu_count() or 2)))
        with concurrent.futures.ThreadPoolExecutor(max_workers=workers) as pool:
            futures = {pool.submit(_run, cmd): label for label, cmd in jobs}
            for fut in concurrent.futures.as_completed(futures):
                fut.result()
                if verbose:
                    print(futures[fut])
    if force or _newer(LIB, objs):
        if verbose:
            print("link", os.path.basename(LIB))
        _run([hipcc, "--offload-arch=" + ARCH, "-shared", "-fPIC", "-o", LIB] + objs + ["-lm", "-lpthread"])
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))

"""Developer tool (GPU box): host-side duration of every call of the end-to-end MPC step loop
(ndlqr_BatchStepAsync + ndlqr_BatchSynchronizePrevious), to see what blocks and what overlaps."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rslqr_amd  # noqa: E402

if "--lib-first" in sys.argv:
    rslqr_amd.lib()
    print("device count", rslqr_amd.device_count())
if "--with-torch" in sys.argv:
    import torch
    torch.cuda.set_device(0)
    torch.cuda.synchronize()
    _t = torch.zeros(16, device="cuda")
    print("torch", torch.__version__, "hip", torch.version.hip, float((_t + 1).sum()))
    import ctypes
    print([l.split()[-1] for l in open("/proc/self/maps") if "libamdhip64" in l][:1])
if "--other-solvers" in sys.argv:  # streams created and destroyed before ours (bench.py's secondary legs)
    for _ in range(3):
        tmp = rslqr_amd.BatchSolver(12, 4, 64, 32)
        tmp.initialize_synthetic(5)
        tmp.solve_async(); tmp.solve_async(); tmp.synchronize()
        if _ < 2:
            tmp.close()
n, m, N, batch = 12, 4, 256, 1024
bs = rslqr_amd.BatchSolver(n, m, N, batch)
bs.initialize_synthetic(1)
bs.solve()
if "--pre-depth" in sys.argv:
    for _ in range(4):
        bs.solve_async()
    bs.synchronize()
    bs.set_pipeline_depth(1)
    for _ in range(4):
        bs.solve_async()
    bs.synchronize()
    bs.set_pipeline_depth(2)
if "--pre-profile" in sys.argv:
    for _ in range(4):
        bs.solve_async()
    bs.synchronize()
    bs.set_flags(rslqr_amd.FLAG_PROFILE)
    bs.solve()
    bs.profile_reset()
    for _ in range(4):
        bs.solve_async()
    bs.synchronize()
    bs.set_flags(0)
    for _ in range(3):
        bs.solve()
if "--pre-kkt" in sys.argv:
    bs.kkt_residuals()
if "--like-bench" in sys.argv:
    sol = bs.solutions()
    pin = rslqr_amd.pinned_empty((batch, bs.nvars))
    bs.solutions(out=pin)
q, r, d, x0 = (rslqr_amd.pinned_empty(s) for s in ((batch, N, n), (batch, N, m), (batch, N, n), (batch, n)))
for a in (q, r, d, x0):
    a[...] = 0.1
outs = [rslqr_amd.pinned_empty((batch, bs.nvars)) for _ in range(2)]
for variant in ("steps", "x0-steps", "solves-only"):
    for rep in range(2):
        t_enq, t_wait = [], []
        t0 = time.perf_counter()
        for i in range(12):
            a = time.perf_counter()
            if variant == "steps":
                bs.step_async(q, r, d, x0, outs[i & 1])
            elif variant == "x0-steps":
                bs.step_async(None, None, None, x0, outs[i & 1])
            else:
                bs.solve_async()
            b = time.perf_counter()
            if i >= 1 and variant != "solves-only":
                bs.synchronize_previous()
            c = time.perf_counter()
            t_enq.append((b - a) * 1e3)
            t_wait.append((c - b) * 1e3)
        bs.synchronize()
        tot = (time.perf_counter() - t0) * 1e3
        print(variant, "total %.3f ms / 12 = %.3f" % (tot, tot / 12))
        print("  enqueue ms:", " ".join("%.3f" % v for v in t_enq))
        print("  wait    ms:", " ".join("%.3f" % v for v in t_wait))

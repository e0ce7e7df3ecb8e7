"""Developer tool (GPU box): the separator-only runtime-sized schedule (generic-reduced) against the knot-based one
(generic-lean, NDLQR_DEV_NO_REDUCED_GENERIC=1) at small batches, ms per step."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CODE = """
import sys, time
sys.path.insert(0, %r)
import numpy as np
import rslqr_amd as R
n, m, N, batch = %d, %d, %d, %d
bs = R.BatchSolver(n, m, N, batch)
bs.initialize_synthetic(3)
bs.solve()
best = 1e9
for _ in range(3):
    t0 = time.perf_counter()
    for _ in range(4):
        bs.solve_async()
    bs.synchronize()
    best = min(best, (time.perf_counter() - t0) / 4 * 1e3)
res, bn = bs.kkt_residuals()
print("%%s %%.3f %%.1e" %% (bs.schedule(), best, (res / np.maximum(1, bn)).max()))
"""
for shape in [(64, 16, 512, 1), (64, 16, 512, 4), (64, 16, 512, 16), (64, 16, 512, 64), (128, 16, 64, 8), (128, 16, 64, 64),
              (96, 16, 256, 4), (32, 8, 256, 4), (32, 8, 256, 64), (48, 16, 128, 8)]:
    out = []
    for env in ({}, {"NDLQR_DEV_NO_REDUCED_GENERIC": "1"}):
        e = dict(os.environ, **env)
        r = subprocess.run([sys.executable, "-c", CODE % ((ROOT,) + shape)], capture_output=True, text=True, env=e)
        out.append(r.stdout.strip().splitlines()[-1] if r.stdout.strip() else "failed: " + r.stderr[-200:])
    print(shape, " | ".join(out), flush=True)

"""Developer tool (GPU box, under rocprofv3 --kernel-trace --stats): the loop that replaces the whole problem from device
memory every iteration -- ndlqr_InitializeBatchFlatDevice + ndlqr_SolveBatchSlicesAsync -- for per-kernel times."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rslqr_amd as R  # noqa: E402

n, m, N, batch = 12, 4, 256, 1024
bs = R.BatchSolver(n, m, N, batch)
bs.initialize_synthetic(1)
gens = [R.generate_synthetic(n, m, N, 1 + p) for p in range(batch)]
dall = [R.DeviceArray((batch,) + gens[0][k].shape).set(np.stack([g[k] for g in gens])) for k in ("A", "B", "Q", "R", "q", "r", "d", "x0")]
du0 = [R.DeviceArray((batch, 1, m)) for _ in range(2)]
for i in range(12):
    bs.initialize_flat_device(*[a.ptr for a in dall])
    assert bs.solve_slices_async(0, 1, R.SOLN_INPUT, du0[i & 1]) == 0
bs.synchronize()
print(bs.schedule())

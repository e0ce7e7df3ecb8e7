"""Developer tool (GPU box): how much do consecutive solves overlap when they do not share buffers?

    python tools/overlap_probe.py

Times K solves of ONE BatchSolver(12,4,256,1024) (stream-ordered: a solve starts when the previous one
has finished) against P solvers of the same size, each on its own stream, solves enqueued round-robin
-- what a double-buffered solver could reach for a stream of independent batches: the thinly populated
upper-level kernels and the HBM-bound back-substitution of one solve run beside the ALU-bound bottom
kernel of the next."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rslqr_amd  # noqa: E402

n, m, N, B, K = 12, 4, 256, 1024, 200


def run(parts):
    solvers = []
    for p in range(parts):
        bs = rslqr_amd.BatchSolver(n, m, N, B)
        bs.initialize_synthetic(1 + p * B)
        solvers.append(bs)
    for _ in range(5):
        for bs in solvers:
            bs.solve_async()
    for bs in solvers:
        bs.synchronize()
    t0 = time.perf_counter()
    for _ in range(K):
        for bs in solvers:
            bs.solve_async()
    for bs in solvers:
        bs.synchronize()
    dt = time.perf_counter() - t0
    for bs in solvers:
        bs.close()
    return dt / (K * parts) * 1e3


for parts in [int(a) for a in sys.argv[1:]] or [1, 2, 3]:
    ms = run(parts)
    print("%d solver(s) in flight: %.4f ms per 1024-batch solve = %.0f solves/s" % (parts, ms, B / ms * 1e3), flush=True)

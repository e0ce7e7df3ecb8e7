"""Developer tool (GPU box): does the back-substitution (HBM bound) of one half-batch overlap with
the factorisation (fp64-issue bound) of the other when the halves run on two streams?

    python tools/overlap_probe.py [halves ...]

Times K solves of one BatchSolver(12,4,256,1024) against P solvers of 1024/P problems each, every
solver on its own stream, solves enqueued round-robin."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rslqr_amd

n, m, N, B, K = 12, 4, 256, 1024, 200


def run(parts):
    solvers = []
    for p in range(parts):
        bs = rslqr_amd.BatchSolver(n, m, N, B // parts)
        bs.initialize_synthetic(1 + p * (B // parts))
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
    return dt / K * 1e3


for parts in [int(a) for a in sys.argv[1:]] or [1, 2, 4]:
    ms = run(parts)
    print("parts %d: %.3f ms per 1024 problems = %.0f solves/s" % (parts, ms, B / ms * 1e3), flush=True)

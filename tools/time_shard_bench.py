"""Developer tool (GPU box): where does cutting ONE problem's horizon over G = 2 ranks pay?

    python tools/time_shard_bench.py            # two rank processes (gloo) on the one GPU of the box + a single-process part

Per horizon N, batch 1, (12,4), medians over repetitions, host clock around synchronised calls:
  single      one GPU, the library's own choice of schedule (tree schedule up to 8192 knots at batch 1)
  single_lvl  one GPU, level-per-launch schedule (NDLQR_TREE=0): the launches the sharded form is made of
  phase0      rank 0 ALONE on the GPU: bottom kernel + the levels inside its chunk of N / 2 knots
  export+import   the top slot out of and back into the device (host staging)
  phase1      the root level (redundant on every rank) + top-down sweep + back-substitution of the chunk
  allreduce   the exchange itself between the two rank processes: gloo over loopback here (host tensors). Two GPUs of
              one node would run it as one RCCL all-reduce of 3.8 KB over xGMI -- latency-bound, ~10-20 us
  sharded_2proc   the whole sharded solve as the two processes run it SIDE BY SIDE ON ONE GPU (they share it: an upper
              bound of what two GPUs would need, minus nothing)
  projected   phase0 + export/import + phase1 + 15 us: the sharded solve with a GPU per rank
"""
import json
import os
import socket
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
NS = [512, 1024, 2048, 4096, 8192]
REPS = 40
BATCH = int(os.environ.get("TSB_BATCH", "1"))  # problems per solve (all of them cut over the two ranks)


def med(f, reps=REPS):
    t = []
    for _ in range(reps):
        t0 = time.perf_counter()
        f()
        t.append((time.perf_counter() - t0) * 1e3)
    return float(np.median(t))


def single_part():
    import rslqr_amd
    out = {}
    for N in NS:
        row = {}
        for name, env in (("single", None), ("single_lvl", {"NDLQR_TREE": "0"})):
            for k, v in (env or {}).items():
                os.environ[k] = v
            bs = rslqr_amd.BatchSolver(12, 4, N, BATCH)
            for k in (env or {}):
                del os.environ[k]
            bs.initialize_synthetic(5)
            bs.set_pipeline_depth(1)
            for _ in range(5):
                bs.solve()
            row[name + "_ms"] = med(bs.solve)
            row[name + "_device_ms"] = bs.solve_ms()
            row[name + "_schedule"] = bs.schedule()
            bs.close()
        bs = rslqr_amd.BatchSolver(12, 4, N, BATCH)
        bs.initialize_synthetic(5)
        cnt = bs.time_shard_top_doubles(2)
        buf = np.zeros(cnt)

        def p0():
            bs.time_shard_factor(0, 2)
            bs.synchronize()

        def xi():
            bs.time_shard_export(2, buf.ctypes.data)
            bs.time_shard_import(2, buf.ctypes.data)

        def p1():
            bs.time_shard_finish(0, 2)
            bs.synchronize()
        for _ in range(3):
            p0(); xi(); p1()
        t0, tx, t1 = [], [], []
        for _ in range(REPS):
            a = time.perf_counter(); p0(); b = time.perf_counter(); xi(); c = time.perf_counter(); p1(); d = time.perf_counter()
            t0.append(b - a); tx.append(c - b); t1.append(d - c)
        row.update(phase0_ms=float(np.median(t0)) * 1e3, export_import_ms=float(np.median(tx)) * 1e3,
                   phase1_ms=float(np.median(t1)) * 1e3, top_slot_bytes=8 * cnt)
        row["projected_two_gpus_ms"] = row["phase0_ms"] + row["export_import_ms"] + row["phase1_ms"] + 0.015
        bs.close()
        out[N] = row
    return out


def rank_main(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    import torch
    import torch.distributed as dist
    import rslqr_amd
    from rslqr_amd import sharding
    dist.init_process_group("gloo", rank=rank, world_size=world)
    out = {}
    for N in NS:
        bs = rslqr_amd.BatchSolver(12, 4, N, BATCH, device=0)
        bs.initialize_synthetic(5)
        for _ in range(5):
            sharding.solve_time_sharded(bs, rank, world)
        dist.barrier()
        ts = med(lambda: sharding.solve_time_sharded(bs, rank, world))
        t = torch.zeros(bs.time_shard_top_doubles(world), dtype=torch.float64)
        dist.barrier()
        ta = med(lambda: dist.all_reduce(t), 200)
        out[N] = {"sharded_2proc_one_gpu_ms": ts, "allreduce_gloo_ms": ta}
        bs.close()
    if rank == 0:
        q.put(out)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    import torch.multiprocessing as mp
    table = single_part()
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=rank_main, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    two = q.get(timeout=600)
    for p in procs:
        p.join()
    for N in NS:
        table[N].update(two[N])
    print(json.dumps({"workload": "(12,4,N) x %d, G = 2" % BATCH, "rows": table}, indent=1))
    print("%6s %9s %11s %9s %9s %9s %11s %13s %11s" % ("N", "single", "single_lvl", "phase0", "exp+imp", "phase1", "allreduce", "sharded(1GPU)", "projected"))
    for N in NS:
        r = table[N]
        print("%6d %9.3f %11.3f %9.3f %9.3f %9.3f %11.3f %13.3f %11.3f   [%s]" % (
            N, r["single_ms"], r["single_lvl_ms"], r["phase0_ms"], r["export_import_ms"], r["phase1_ms"],
            r["allreduce_gloo_ms"], r["sharded_2proc_one_gpu_ms"], r["projected_two_gpus_ms"], r["single_schedule"]))

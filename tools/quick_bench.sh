#!/bin/bash
# Developer tool (GPU box): headline step + per-kernel HIP-event times, three runs.   bash tools/quick_bench.sh [tag] [bench args]
tag=${1:-q}; shift || true
for i in 1 2 3; do
python3 bench.py --steps ${STEPS:-100} --warmup 5 --no-configs --no-transfers --no-cpu --no-modes "$@" > gpurun_out/${tag}_$i.json 2> gpurun_out/${tag}_$i.err
python3 - <<PY
import json
d=json.load(open("gpurun_out/${tag}_$i.json")); r=d["roofline"]
ks={r["kernel"]:(r["avg_launch_ms"],r["launches_per_step"])}; ks.update({k:(v["avg_launch_ms"],v["launches_per_step"]) for k,v in r["kernels"].items()})
print("${tag} run $i: %.0f solves/s  %.4f ms/step  depth1 %.4f | "%(d["value"],d["ms_per_step"],d["pipeline"]["ms_per_step_depth1"])+"  ".join("%s %.1f us x%d"%(k,v[0]*1e3,v[1]) for k,v in ks.items()), "| kkt %.1e"%d["config"]["kkt_residual_rel_max"])
PY
done

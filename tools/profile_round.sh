#!/bin/bash
# Developer tool, run on the GPU box:  bash tools/profile_round.sh TAG [bench.py workload args]
#   e.g. bash tools/profile_round.sh r02c        (headline workload)
#        bash tools/profile_round.sh r02_config5 --nx 64 --nu 16 --horizon 512 --batch 256
# 1. bench.py (with the cpu_baseline leg)          -> gpurun_out/TAG_bench.json
# 2. rocprofv3 --kernel-trace --stats of bench.py   -> gpurun_out/TAG_stats/
# 3. rocprofv3 --pmc FETCH_SIZE, WRITE_SIZE (separate passes, no trace domains)
#                                                   -> gpurun_out/TAG_pmc_{fetch,write}/
# 4. summaries: gpurun_out/TAG_kernel_stats.csv, TAG_pmc.txt, TAG_traffic.json
# Copy what should be judged into profiles/ afterwards.
set -e
tag=${1:-rXX}
shift || true
root=$(pwd)
export TMPDIR=/tmp
python3 bench.py --steps 20 --warmup 5 "$@" > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err
echo "[profile_round] bench done"
# the profiler passes run the solves strictly stream-ordered: per-kernel durations then are those of the
# kernel alone (what bench.py's `roofline.kernels` holds), not of two pipelined solves sharing the chip
export NDLQR_PIPELINE=1
rocprofv3 --kernel-trace --stats --output-format csv -d $root/gpurun_out/${tag}_stats -o run -- python3 bench.py --no-cpu --no-modes --no-configs --no-transfers "$@" \
    > gpurun_out/${tag}_bench_under_rocprof.json 2> gpurun_out/${tag}_stats.err
echo "[profile_round] kernel trace done"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $root/gpurun_out/${tag}_pmc_fetch -o run -- python3 bench.py --no-cpu --no-modes --no-configs --no-transfers --steps 2 --warmup 1 "$@" \
    > /dev/null 2> gpurun_out/${tag}_pmc_fetch.err
echo "[profile_round] FETCH_SIZE pass done"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $root/gpurun_out/${tag}_pmc_write -o run -- python3 bench.py --no-cpu --no-modes --no-configs --no-transfers --steps 2 --warmup 1 "$@" \
    > /dev/null 2> gpurun_out/${tag}_pmc_write.err
echo "[profile_round] WRITE_SIZE pass done"
cp $(find gpurun_out/${tag}_stats -name "*kernel_stats.csv" | head -1) gpurun_out/${tag}_kernel_stats.csv
f=$(find gpurun_out/${tag}_pmc_fetch -name "*counter_collection.csv" | head -1)
w=$(find gpurun_out/${tag}_pmc_write -name "*counter_collection.csv" | head -1)
python3 tools/pmc_summary.py $f $w > gpurun_out/${tag}_pmc.txt
# workload of the run, from the bench line itself
read nx nu N batch flags < <(python3 - <<PY
import json, re
d = json.load(open("gpurun_out/${tag}_bench.json"))
m = re.search(r"nx=(\d+) nu=(\d+) N=(\d+) batch=(\d+)", d["config"]["workload"])
print(*m.groups(), d["config"]["flags"])
PY
)
python3 tools/make_traffic.py $tag $f $w $nx $nu $N $batch $flags > gpurun_out/${tag}_traffic.json
cat gpurun_out/${tag}_pmc.txt
head -8 gpurun_out/${tag}_kernel_stats.csv | cut -c1-200

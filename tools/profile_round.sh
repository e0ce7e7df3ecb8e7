#!/bin/bash
# Developer tool, run on the GPU box:  bash tools/profile_round.sh TAG
# 1. bench.py (with the cpu_baseline leg)          -> gpurun_out/TAG_bench.json
# 2. rocprofv3 --kernel-trace --stats of bench.py   -> gpurun_out/TAG_stats/
# 3. rocprofv3 --pmc FETCH_SIZE, WRITE_SIZE (separate passes, no trace domains)
#                                                   -> gpurun_out/TAG_pmc_{fetch,write}/
# Copy what should be judged into profiles/ afterwards (tools/pmc_summary.py for the PMC pair).
set -e
tag=${1:-rXX}
root=$(pwd)
export TMPDIR=/tmp
python3 bench.py > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err
echo "[profile_round] bench done"
rocprofv3 --kernel-trace --stats --output-format csv -d $root/gpurun_out/${tag}_stats -o run -- python3 bench.py --no-cpu \
    > gpurun_out/${tag}_bench_under_rocprof.json 2> gpurun_out/${tag}_stats.err
echo "[profile_round] kernel trace done"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $root/gpurun_out/${tag}_pmc_fetch -o run -- python3 bench.py --no-cpu --steps 2 --warmup 1 \
    > /dev/null 2> gpurun_out/${tag}_pmc_fetch.err
echo "[profile_round] FETCH_SIZE pass done"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $root/gpurun_out/${tag}_pmc_write -o run -- python3 bench.py --no-cpu --steps 2 --warmup 1 \
    > /dev/null 2> gpurun_out/${tag}_pmc_write.err
echo "[profile_round] WRITE_SIZE pass done"
find gpurun_out/${tag}_stats gpurun_out/${tag}_pmc_fetch gpurun_out/${tag}_pmc_write -name "*.csv" | head -20

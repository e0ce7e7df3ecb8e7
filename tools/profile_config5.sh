#!/bin/bash
# Developer tool, GPU box: kernel trace + MFMA / HBM counters of the (64,16,512) large-block path.
#   bash tools/profile_config5.sh TAG
set -e
tag=${1:-rXX}
root=$(pwd)
export TMPDIR=/tmp
python3 tools/bench_config5.py 64 > gpurun_out/${tag}_config5_batch64.txt 2>/dev/null
python3 tools/bench_config5.py 256 >> gpurun_out/${tag}_config5_batch64.txt 2>/dev/null || true
rocprofv3 --kernel-trace --stats --output-format csv -d $root/gpurun_out/${tag}_c5_stats -o run -- python3 tools/bench_config5.py 16 \
    > gpurun_out/${tag}_c5_stats.txt 2> gpurun_out/${tag}_c5_stats.err
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES --output-format csv -d $root/gpurun_out/${tag}_c5_mfma -o run -- python3 tools/bench_config5.py 16 \
    > /dev/null 2> gpurun_out/${tag}_c5_mfma.err
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $root/gpurun_out/${tag}_c5_fetch -o run -- python3 tools/bench_config5.py 16 > /dev/null 2> gpurun_out/${tag}_c5_fetch.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $root/gpurun_out/${tag}_c5_write -o run -- python3 tools/bench_config5.py 16 > /dev/null 2> gpurun_out/${tag}_c5_write.err
cat gpurun_out/${tag}_config5_batch64.txt

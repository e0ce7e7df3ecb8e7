#!/bin/bash
# Developer tool (GPU box): per-kernel durations of a solve mode on the headline shape (rocprofv3 kernel trace).
#   bash tools/mode_trace.sh FLAGS [n m N batch]      FLAGS: 8 = KEEP_FACT, 1 = STRICT_FP, 16 = KEEP_RECORDS, 0 = default
export TMPDIR=/tmp NDLQR_PIPELINE=1
root=$(pwd)
flags=${1:-8}; n=${2:-12}; m=${3:-4}; N=${4:-256}; batch=${5:-1024}
cat > /tmp/mode_trace.py <<PY
import sys
sys.path.insert(0, "$root")
import rslqr_amd as R
bs = R.BatchSolver($n, $m, $N, $batch, flags=$flags)
bs.initialize_synthetic(1)
for _ in range(8):
    bs.solve_async()
bs.synchronize()
print(bs.schedule())
PY
rocprofv3 --kernel-trace --stats --output-format csv -d $root/gpurun_out/mode_trace_$flags -o run -- python3 /tmp/mode_trace.py > gpurun_out/mode_trace_$flags.txt 2>&1
tail -1 gpurun_out/mode_trace_$flags.txt
python3 tools/kstats.py $(find gpurun_out/mode_trace_$flags -name "*kernel_stats.csv" | head -1) | head -12

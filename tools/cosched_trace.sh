#!/bin/bash
# Developer tool (GPU box): per-kernel durations of the co-scheduled form against the default (rocprofv3 kernel trace of
# tools/cosched_probe.py --time-only on the headline shape).   bash tools/cosched_trace.sh
export TMPDIR=/tmp
root=$(pwd)
for cos in 0 1; do
  NDLQR_PROBE_COSCHED=$cos NDLQR_PROBE_SHAPES=${SHAPES:-12,4,256,1024} rocprofv3 --kernel-trace --stats --output-format csv \
      -d $root/gpurun_out/cosched_trace_$cos -o run -- python3 tools/cosched_probe.py --time-only > gpurun_out/cosched_trace_$cos.txt 2>&1
  echo "== NDLQR_COSCHED=$cos"; cat gpurun_out/cosched_trace_$cos.txt | grep ms/step
  python3 tools/kstats.py $(find gpurun_out/cosched_trace_$cos -name "*kernel_stats.csv" | head -1) | head -8
done

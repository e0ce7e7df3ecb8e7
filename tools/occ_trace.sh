#!/bin/bash
# Developer tool (GPU box): bottom kernel at reduced occupancy -- libraries built with -DNDLQR_DEV_LDS_PAD=<bytes>
# (rslqr_amd/librslqr_amd_pad<bytes>.so, loaded through NDLQR_LIBRARY) --, LDS block size and duration per kernel from the kernel trace.
export TMPDIR=/tmp
root=$(pwd)
for pad in ${1:-0 11000 30000}; do
  if [ $pad = 0 ]; then unset NDLQR_LIBRARY; else export NDLQR_LIBRARY=$root/rslqr_amd/librslqr_amd_pad$pad.so; fi
  export NDLQR_PIPELINE=1
  rocprofv3 --kernel-trace --output-format csv -d $root/gpurun_out/occ_trace_${pad} -o run -- python3 bench.py --no-cpu --no-modes --no-configs --no-transfers --steps 20 > /dev/null 2>&1
  python3 - <<PY
import csv, glob, collections
f = glob.glob("gpurun_out/occ_trace_${pad}/**/*kernel_trace.csv", recursive=True)[0]
acc = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    n = r["Kernel_Name"].split("(")[0].replace("void ndlqr::", "")
    acc[(n[:40], r["LDS_Block_Size"], r["VGPR_Count"], r["Workgroup_Size_X"])].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
for k, v in sorted(acc.items(), key=lambda kv: -sum(kv[1]))[:5]:
    v.sort()
    print("pad $pad", k, "calls", len(v), "median %.1f us" % (v[len(v) // 2] / 1e3))
PY
done

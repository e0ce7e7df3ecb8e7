#!/bin/bash
# Developer tool (GPU box): kernel + memory-copy timeline of the end-to-end step loop (tools/e2e_probe.py)
export TMPDIR=/tmp
root=$(pwd)
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $root/gpurun_out/e2e_trace -o run -- python3 tools/e2e_probe.py > gpurun_out/e2e_trace.log 2>&1
python3 - <<PY
import csv, glob
rows = []
for f in glob.glob("gpurun_out/e2e_trace/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:40], r.get("Stream_Id", r.get("Queue_Id", "?"))))
for f in glob.glob("gpurun_out/e2e_trace/**/*memory_copy_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "COPY " + r.get("Direction", "?"), r.get("Stream_Id", "?")))
rows.sort()
# the last 60 events of the first variant's steady state: find pack_rhs_stream launches
idx = [i for i, r in enumerate(rows) if "pack_rhs_stream" in r[2]]
if idx:
    lo = idx[min(len(idx) - 1, 14)]
    t0 = rows[lo][0]
    for r in rows[lo:lo + 45]:
        print("%9.3f %9.3f  %-42s q=%s" % ((r[0] - t0) / 1e3, (r[1] - t0) / 1e3, r[2], r[3]))
PY

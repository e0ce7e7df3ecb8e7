#!/bin/bash
# Developer tool (GPU box): clock and issue counters of the co-scheduled launch against its two parts
# (GRBM_GUI_ACTIVE / duration = effective clock).   bash tools/cosched_pmc.sh
export TMPDIR=/tmp
root=$(pwd)
for cos in 0 1; do
  if [ $cos = 0 ]; then export NDLQR_PIPELINE=1; else unset NDLQR_PIPELINE; fi
  NDLQR_COSCHED=$cos rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU --output-format csv -d $root/gpurun_out/cosched_pmc_$cos -o run -- python3 bench.py --no-cpu --no-modes --no-configs --no-transfers --spin-up-ms 0 --steps 6 --warmup 2 > /dev/null 2> gpurun_out/cosched_pmc_$cos.err
  python3 - <<PY
import csv, glob, collections
f = glob.glob("gpurun_out/cosched_pmc_$cos/**/*counter_collection.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
print(list(rows[0].keys()))
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    n = r["Kernel_Name"].split("(")[0].replace("void ndlqr::", "")[:36] + " g" + r.get("Grid_Size", "?")
    acc[n][r["Counter_Name"]].append(float(r["Counter_Value"]))
    if "Start_Timestamp" in r:
        acc[n]["dur_ns"].append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
for n, cs in acc.items():
    if "mc" not in n and "backsub" not in n: continue
    print("cosched $cos", n, {c: "%.4g" % (sum(v) / len(v)) for c, v in sorted(cs.items())}, "launches", len(cs["GRBM_GUI_ACTIVE"]))
PY
done

#!/bin/bash
# Developer tool (GPU box): SQ counters of the bottom kernel, shipped library against rslqr_amd/librslqr_amd_<tag>.so.
tag=$1; shift
export TMPDIR=/tmp NDLQR_PIPELINE=1
root=$(pwd)
for lib in shipped $tag; do
  if [ $lib = shipped ]; then unset NDLQR_LIBRARY; else export NDLQR_LIBRARY=$root/rslqr_amd/librslqr_amd_$lib.so; fi
  i=0
  for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_INSTS_MFMA" \
             "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS"; do
    rocprofv3 --pmc $set --output-format csv -d $root/gpurun_out/sqab_${lib}_$i -o run -- python3 bench.py --no-cpu --no-modes --no-configs --no-transfers --spin-up-ms 0 --steps 2 --warmup 1 "$@" > /dev/null 2> gpurun_out/sqab_${lib}_$i.err
    i=$((i+1))
  done
  python3 - <<PY
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob("gpurun_out/sqab_${lib}_*/**/*counter_collection.csv", recursive=True)):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"].split("(")[0].replace("void ndlqr::", "")[:40] + " g" + row.get("Grid_Size", "?")
        acc[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k, cs in acc.items():
    if "mc<" not in k and "backsub" not in k: continue
    w = sum(cs["SQ_WAVES"]) / len(cs["SQ_WAVES"])
    print("$lib", k, " ".join("%s %.4g" % (c.replace("SQ_", ""), sum(v) / len(v) / (w if c != "SQ_WAVES" else 1)) for c, v in sorted(cs.items())))
PY
done

#!/bin/bash
# Developer tool, run on the GPU box:  bash tools/sq_counters.sh TAG [bench.py workload args]
# SQ counters of the default workload in separate --pmc passes (no trace domains) -> gpurun_out/TAG_sq_*/
tag=${1:-rXX}
shift || true
root=$(pwd)
export TMPDIR=/tmp
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_MFMA" \
           "SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_CYCLES_VMEM SQ_WAIT_ANY SQ_WAIT_INST_ANY" \
           "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_MISC SQ_INSTS_SMEM SQ_INSTS_VMEM"; do
  rocprofv3 --pmc $set --output-format csv -d $root/gpurun_out/${tag}_sq_$i -o run -- python3 bench.py --no-cpu --no-modes --no-configs --no-transfers --spin-up-ms 0 --steps 2 --warmup 1 "$@" \
      > /dev/null 2> gpurun_out/${tag}_sq_$i.err || echo "pass $i failed"
  echo "[sq_counters] pass $i done"
  i=$((i+1))
done
python3 - <<PY
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob("gpurun_out/${tag}_sq_*/**/*counter_collection.csv", recursive=True)):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"].split("(")[0][:60] + "  grid " + row.get("Grid_Size", "?")  # (per tree level)
        acc[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
with open("gpurun_out/${tag}_sq_summary.txt", "w") as out:
    for k, cs in acc.items():
        if "ndlqr" not in k: continue
        out.write(k + "  (per launch averages)\n")
        for c, v in sorted(cs.items()):
            out.write("    %-28s %.4g  (%d launches)\n" % (c, sum(v) / len(v), len(v)))
print(open("gpurun_out/${tag}_sq_summary.txt").read())
PY
# instructions per wavefront and the fp64-pipe time of every kernel kind -> gpurun_out/TAG_issue.json (needs the workload:
# default = the headline one)
read nx nu N batch < <(python3 - "$@" <<PY
import sys
a = sys.argv[1:]
g = lambda k, d: a[a.index(k) + 1] if k in a else d
print(g("--nx", 12), g("--nu", 4), g("--horizon", 256), g("--batch", 1024))
PY
)
python3 tools/make_issue.py $tag $nx $nu $N $batch 0 > gpurun_out/${tag}_issue.json


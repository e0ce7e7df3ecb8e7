#!/bin/bash
# Developer tool: install the summaries of a profile round (tools/profile_round.sh + tools/sq_counters.sh, headline and config 5)
# from gpurun_out/ into profiles/ under TAG, drop the set OLDTAG and rename its mentions in the documents.
#   bash tools/install_profiles.sh r04k r04j
set -e
tag=$1; old=$2
for s in "" _config5; do
  for f in bench.json bench_under_rocprof.json kernel_stats.csv pmc.txt traffic.json issue.json; do cp gpurun_out/${tag}${s}_$f profiles/${tag}${s}_$f; done
  cp gpurun_out/${tag}${s}_sq_summary.txt profiles/${tag}${s}_sq_counters.txt
done
if [ -n "$old" ]; then
  git rm -q profiles/${old}_* || true
  sed -i "s/${old}_/${tag}_/g; s/\`${old}\`/\`${tag}\`/g" DESIGN.md README.md profiles/README.md profiles/r04_fuse2_ab.txt
fi
python3 - <<PY
import json, sys
sys.path.insert(0, ".")
import bench
sha = bench.csrc_sha()
for t in ("$tag", "${tag}_config5"):
    for k in ("traffic", "issue"):
        got = json.load(open("profiles/%s_%s.json" % (t, k)))["csrc_sha"]
        print(t, k, got, "ok" if got == sha else "STALE (tree is %s)" % sha)
PY

#!/bin/bash
# Developer tool (GPU box): bottom kernel of the headline shape at reduced occupancy (NDLQR_DEV_LDS_PAD: dynamic LDS
# added to its launch), one solve in flight, per-kernel HIP-event times.   bash tools/occ_probe.sh "0 4680 11000"
export NDLQR_PIPELINE=1
for pad in ${1:-0 4680 11000 30000}; do
  echo "== pad $pad"
  NDLQR_DEV_LDS_PAD=$pad python3 bench.py --no-cpu --no-modes --no-configs --no-transfers --steps 50 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); ks=dict(d['roofline']['kernels']); ks[d['roofline']['kernel']]=d['roofline']
print(round(d['value']), round(d['ms_per_step'],4), {k:round(v['ms_per_step'],4) for k,v in sorted(ks.items())})"
done

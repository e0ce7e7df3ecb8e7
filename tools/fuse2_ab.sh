#!/bin/bash
# Developer tool (GPU box): same-build A/B of the fused bottom launch (NDLQR_FUSE2=1: levels 0-2 in bottom8_reduced_mc) against
# the default (four-knot kernel + a level-2 launch).   bash tools/fuse2_ab.sh [bench args]
for i in 1 2 3; do for f in 0 1; do
  NDLQR_FUSE2=$f python3 bench.py --no-cpu --no-modes --no-configs --no-transfers --steps 100 "$@" 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; ks=dict(r['kernels']); ks[r['kernel']]=r
print('fuse2=$f', d['config']['schedule'], round(d['value']), round(d['ms_per_step'],4), round(d['pipeline']['ms_per_step_depth1'],4), {k:(round(v['avg_launch_ms']*1e3,1), v['launches_per_step']) for k,v in sorted(ks.items())}, 'model GB/step %.3f' % (r['step']['algorithmic_bytes']/1e9))"
done; done

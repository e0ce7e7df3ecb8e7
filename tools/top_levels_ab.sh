#!/bin/bash
# Developer tool (GPU box): same-build A/B of the number of tree levels inside reduced_top_mc (NDLQR_TOP_LEVELS=3 / 4 / 5;
# beyond three a wavefront takes several separators of the launch's first levels in turn).   bash tools/top_levels_ab.sh [bench args]
for i in 1 2 3; do for t in 3 4 5; do
  NDLQR_TOP_LEVELS=$t python3 bench.py --no-cpu --no-modes --no-configs --no-transfers --steps 100 "$@" 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; ks=dict(r['kernels']); ks[r['kernel']]=r
print('top_levels=$t', d['config']['schedule'], round(d['value']), round(d['ms_per_step'],4), round(d['pipeline']['ms_per_step_depth1'],4), {k:(round(v['avg_launch_ms']*1e3,1), v['launches_per_step']) for k,v in sorted(ks.items())}, 'kkt %.1e' % d.get('kkt_residual_rel_max', -1))"
done; done

#!/bin/bash
# Developer tool (GPU box): workgroup size of backsub_multipliers_compact (NDLQR_MULT_THREADS=64 / 128 / 256) per block size
run() { python bench.py --no-cpu --no-modes --no-configs --no-transfers "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; ks=dict(r['kernels']); ks[r['kernel']]=r
print('threads=$NDLQR_MULT_THREADS', d['config']['workload'][:34], '| %.0f solves/s, %.4f ms/step |' % (d['value'], d['ms_per_step']), {k: round(v['ms_per_step'],4) for k,v in ks.items()}, flush=True)"; }
for s in "16 4 256 1024" "20 20 256 256" "32 8 256 256" "48 16 512 256" "64 16 512 256" "96 16 256 64"; do
  set -- $s
  for t in 64 128 256; do export NDLQR_MULT_THREADS=$t; run --nx $1 --nu $2 --horizon $3 --batch $4; done
done

#!/bin/bash
# Developer tool (GPU box): same-box A/B of the shipped library against another build of it (rslqr_amd/librslqr_amd_<tag>.so,
# loaded through NDLQR_LIBRARY), one solve in flight, per-kernel HIP-event times.   bash tools/ab_lib.sh TAG [bench args]
tag=$1; shift
export NDLQR_PIPELINE=1
for i in 1 2; do
for lib in shipped $tag; do
  if [ $lib = shipped ]; then unset NDLQR_LIBRARY; else export NDLQR_LIBRARY=$PWD/rslqr_amd/librslqr_amd_$lib.so; fi
  python3 bench.py --no-cpu --no-modes --no-configs --no-transfers --steps ${STEPS:-50} "$@" 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); ks=dict(d['roofline']['kernels']); ks[d['roofline']['kernel']]=d['roofline']
print('$lib', round(d['value']), round(d['ms_per_step'],4), {k:round(v['ms_per_step'],4) for k,v in sorted(ks.items())}, 'kkt %.1e' % d['config']['kkt_residual_rel_max'])"
done; done

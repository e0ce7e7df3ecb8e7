for i in 1 2 3; do
for lib in prev new; do
  if [ $lib = prev ]; then export NDLQR_LIBRARY=$PWD/rslqr_amd/librslqr_amd_prev.so; else unset NDLQR_LIBRARY; fi
  python bench.py --no-cpu --no-modes --no-configs --no-transfers --steps 40 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); ks=dict(d['roofline']['kernels']); ks[d['roofline']['kernel']]=d['roofline']
print('$lib', round(d['value']), round(d['ms_per_step'],4), round(d['pipeline']['ms_per_step_depth1'],4), {k:round(v['ms_per_step'],4) for k,v in sorted(ks.items())})"
done; done

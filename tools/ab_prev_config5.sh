# Developer tool (GPU box): same-box A/B of two builds of the library on BASELINE config 5, (64,16,512) x 256 -- see
# ab_prev.sh for how to put the other build at rslqr_amd/librslqr_amd_prev.so. Extra arguments go to bench.py.
for i in 1 2; do
for lib in prev new; do
  if [ $lib = prev ]; then export NDLQR_LIBRARY=$PWD/rslqr_amd/librslqr_amd_prev.so; else unset NDLQR_LIBRARY; fi
  python bench.py --nx 64 --nu 16 --horizon 512 --batch 256 --no-cpu --no-modes --no-configs --no-transfers --steps 20 "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); ks=dict(d['roofline']['kernels']); ks[d['roofline']['kernel']]=d['roofline']
print('$lib', round(d['value']), round(d['ms_per_step'],4), {k:(round(v['ms_per_step'],4), v['launches_per_step']) for k,v in sorted(ks.items())}, d['config'].get('kkt_residual_rel_max'))"
done; done

# Developer tool (GPU box): same-box A/B of two builds of the library on the headline workload. Put the OTHER build at
# rslqr_amd/librslqr_amd_prev.so first (e.g. `git stash; python -m rslqr_amd.build; cp rslqr_amd/librslqr_amd.so
# /tmp/prev.so; git stash pop; python -m rslqr_amd.build; cp /tmp/prev.so rslqr_amd/librslqr_amd_prev.so`); the ctypes
# mirror loads it through NDLQR_LIBRARY. Boxes differ by +-4 %: only runs of one gpurun call compare.
for i in 1 2 3; do
for lib in prev new; do
  if [ $lib = prev ]; then export NDLQR_LIBRARY=$PWD/rslqr_amd/librslqr_amd_prev.so; else unset NDLQR_LIBRARY; fi
  python bench.py --no-cpu --no-modes --no-configs --no-transfers --steps ${STEPS:-100} 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); ks=dict(d['roofline']['kernels']); ks[d['roofline']['kernel']]=d['roofline']
print('$lib', round(d['value']), round(d['ms_per_step'],4), round(d['pipeline']['ms_per_step_depth1'],4), {k:round(v['ms_per_step'],4) for k,v in sorted(ks.items())})"
done; done

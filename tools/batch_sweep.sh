# Developer tool (GPU box): per-kernel time per 1024 problems as a function of the batch (does a batch tile whose
# working set -- ~1.2 MB per (12,4,256) problem: inputs, records, slots, solution -- fits the 256 MiB Infinity Cache
# run faster per problem? NDLQR_PIPELINE=1: one solve in flight, so a solve's own footprint is all that is live)
export NDLQR_PIPELINE=1
for b in 64 128 192 256 512 1024 2048; do
python bench.py --no-cpu --no-modes --no-configs --no-transfers --steps 40 --batch $b 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; ks=dict(r['kernels']); ks[r['kernel']]=r
b=$b
print('batch %5d | %s | %.4f ms/step = %.4f ms per 1024 problems |' % (b, d['config']['schedule'], d['ms_per_step'], d['ms_per_step']*1024/b), {k: round(v['ms_per_step']*1024/b,4) for k,v in sorted(ks.items())})"
done

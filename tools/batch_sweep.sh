# per-kernel times of the headline shape against the batch size (developer tool, GPU box):
# does a batch whose working set fits the 256 MiB Infinity Cache run faster per problem?
run() { python bench.py --no-cpu --no-modes "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; ks=dict(r['kernels']); ks[r['kernel']]=r
b=d['config']['batch_per_gpu'] if 'batch_per_gpu' in d['config'] else 0
print(d['config']['workload'], '| %.0f solves/s, %.4f ms/step | depth1 %.0f |' % (d['value'], d['ms_per_step'], d['pipeline']['value_depth1']), {k: round(v['ms_per_step'],4) for k,v in ks.items()})"; }
for b in 64 128 192 256 384 512 1024 2048; do run --nx 12 --nu 4 --horizon 256 --batch $b --steps 100; done

# bench.py --no-cpu over the other BASELINE shapes (developer tool, GPU box): one line each
run() { python bench.py --no-cpu --no-modes "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; ks=dict(r['kernels']); ks[r['kernel']]=r
print(d['config']['workload'], '|', d['config']['schedule'], '| %.0f solves/s, %.4f ms/step |' % (d['value'], d['ms_per_step']), {k: round(v['ms_per_step'],4) for k,v in ks.items()})"; }
run --nx 12 --nu 4 --horizon 1024 --batch 512
run --nx 6 --nu 3 --horizon 256 --batch 1024
run --nx 6 --nu 3 --horizon 256 --batch 1 --steps 200
run --nx 12 --nu 4 --horizon 256 --batch 1 --steps 200
run --nx 12 --nu 4 --horizon 256 --batch 16 --steps 200
run --nx 4 --nu 2 --horizon 256 --batch 1024
run --nx 8 --nu 4 --horizon 256 --batch 1024
run --nx 13 --nu 4 --horizon 256 --batch 1024
run --nx 12 --nu 4 --horizon 256 --batch 1024 --flags 1
run --nx 12 --nu 4 --horizon 256 --batch 1024 --flags 8

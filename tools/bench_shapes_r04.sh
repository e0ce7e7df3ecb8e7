# bench.py --no-cpu over the size-specialised and padded shapes x 1024 and a few runtime-sized ones (developer tool, GPU box)
run() { python bench.py --no-cpu --no-modes --no-configs --no-transfers "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; ks=dict(r['kernels']); ks[r['kernel']]=r
print(d['config']['workload'], '|', d['config']['schedule'], '| %.0f solves/s, %.4f ms/step |' % (d['value'], d['ms_per_step']), {k: round(v['ms_per_step'],4) for k,v in ks.items()}, flush=True)"; }
for s in "6 3" "8 4" "9 3" "10 4" "12 4" "13 4" "12 8" "15 2" "8 16" "4 2" "7 9" "5 3" "11 3" "14 2"; do set -- $s; run --nx $1 --nu $2 --horizon 256 --batch 1024; done
run --nx 12 --nu 4 --horizon 1024 --batch 512
run --nx 12 --nu 4 --horizon 64 --batch 4096
run --nx 6 --nu 3 --horizon 256 --batch 1 --steps 200
run --nx 12 --nu 4 --horizon 256 --batch 1 --steps 200
run --nx 12 --nu 4 --horizon 256 --batch 16 --steps 200
run --nx 16 --nu 4 --horizon 256 --batch 1024
run --nx 20 --nu 20 --horizon 256 --batch 256
run --nx 48 --nu 16 --horizon 512 --batch 256
run --nx 96 --nu 16 --horizon 256 --batch 64
run --nx 128 --nu 16 --horizon 256 --batch 64

run() { python bench.py --no-cpu --no-modes --no-configs --no-transfers --steps 20 "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; ks=dict(r['kernels']); ks[r['kernel']]=r
print(d['config']['workload'], '|', d['config']['schedule'], '| %.0f solves/s, %.4f ms/step |' % (d['value'], d['ms_per_step']), {k: round(v['ms_per_step'],4) for k,v in ks.items()})"; }
echo "rowbcast=1 (12,4)"; NDLQR_ROWBCAST=1 run --nx 12 --nu 4 --horizon 256 --batch 1024
echo "rowbcast=0 (10,4)"; NDLQR_ROWBCAST=0 run --nx 10 --nu 4 --horizon 256 --batch 1024
echo "default (10,4)"; run --nx 10 --nu 4 --horizon 256 --batch 1024
echo "rowbcast=0 (8,4)"; NDLQR_ROWBCAST=0 run --nx 8 --nu 4 --horizon 256 --batch 1024
echo "default (8,4)"; run --nx 8 --nu 4 --horizon 256 --batch 1024
echo "rowbcast=0 (6,3)"; NDLQR_ROWBCAST=0 run --nx 6 --nu 3 --horizon 256 --batch 1024
echo "default (6,3)"; run --nx 6 --nu 3 --horizon 256 --batch 1024
echo "default (13,4)"; run --nx 13 --nu 4 --horizon 256 --batch 1024
echo "default (9,3)"; run --nx 9 --nu 3 --horizon 256 --batch 1024

# Developer tool (GPU box): block sizes without a size-specialised instance, zero-padded into the next instance
# (default) against the runtime-sized kernels on their own block size (NDLQR_NO_PAD=1)
run() { python bench.py --no-cpu --no-modes --no-configs --no-transfers --steps 20 "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; ks=dict(r['kernels']); ks[r['kernel']]=r
print(d['config']['workload'], '|', d['config']['schedule'], '| %.0f solves/s, %.4f ms/step | kkt %.1e |' % (d['value'], d['ms_per_step'], d['config']['kkt_residual_rel_max']), {k: round(v['ms_per_step'],4) for k,v in ks.items()})"; }
for shape in "7 9 256 1024" "5 3 256 1024" "11 3 256 1024" "14 2 256 1024" "3 1 256 1024" "9 6 256 1024" "7 2 256 1024"; do
  set -- $shape
  echo "padded:"; run --nx $1 --nu $2 --horizon $3 --batch $4
  echo "own size:"; NDLQR_NO_PAD=1 run --nx $1 --nu $2 --horizon $3 --batch $4
done

# developer tool (GPU box): block sizes without a size-specialised instance -- the separator-only schedule on the
# matrix cores (zero-padded tiles where the block does not fill them) against the knot-based runtime-sized kernels
run() { python bench.py --no-cpu --no-modes "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; ks=dict(r['kernels']); ks[r['kernel']]=r
print('   %-16s %9.0f solves/s  %8.4f ms/step ' % (d['config']['schedule'], d['value'], d['ms_per_step']), {k: round(v['ms_per_step'],3) for k,v in ks.items()})"; }
for shape in "20 20 256 256" "24 6 256 512" "32 8 256 512" "16 4 256 1024" "7 9 256 1024" "48 16 512 256"; do set -- $shape
  echo "nx=$1 nu=$2 N=$3 batch=$4"
  run --nx $1 --nu $2 --horizon $3 --batch $4 --steps 20
  NDLQR_NO_MFMA=1 run --nx $1 --nu $2 --horizon $3 --batch $4 --steps 20
done

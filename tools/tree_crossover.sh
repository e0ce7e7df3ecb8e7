# developer tool (GPU box): tree schedule (one launch for the whole factorisation) against one launch per level,
# by batch size -- where plan_small's threshold belongs
run() { python bench.py --no-cpu --no-modes "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('   %-14s %8.0f solves/s  %.4f ms/step  depth1 %.4f ms' % (d['config']['schedule'], d['value'], d['ms_per_step'], d['pipeline']['ms_per_step_depth1']))"; }
for shape in "12 4" "6 3"; do set -- $shape
  for b in 8 16 32 64 128 256; do
    echo "nx=$1 nu=$2 N=256 batch=$b"
    NDLQR_TREE=1 run --nx $1 --nu $2 --horizon 256 --batch $b --steps 200
    NDLQR_TREE=0 run --nx $1 --nu $2 --horizon 256 --batch $b --steps 200
  done
done

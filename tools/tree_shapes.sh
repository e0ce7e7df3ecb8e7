run() { python bench.py --no-cpu --no-modes "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print(d['config']['workload'], '|', d['config']['schedule'], '| %.0f solves/s, %.4f ms/step | depth1 %.4f ms' % (d['value'], d['ms_per_step'], d['pipeline']['ms_per_step_depth1']), 'device median %.4f' % d['device_ms_per_step']['median'])"; }
run --nx 6 --nu 3 --horizon 256 --batch 1 --steps 300
run --nx 12 --nu 4 --horizon 256 --batch 1 --steps 300
run --nx 12 --nu 4 --horizon 256 --batch 16 --steps 300
run --nx 12 --nu 4 --horizon 256 --batch 32 --steps 300

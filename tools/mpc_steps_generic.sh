#!/bin/bash
# Developer tool (GPU box): MPC step with and without kept records on the runtime-sized separator-only schedule (where does the
# re-solve pay?)
for s in "20 20 256 256" "24 6 256 512" "32 8 256 256" "48 16 512 256"; do
  set -- $s
  python bench.py --no-cpu --no-configs --no-modes --nx $1 --nu $2 --horizon $3 --batch $4 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); e=d['end_to_end']
print('($1,$2,$3) x $4  solve %.4f ms |' % d['ms_per_step'], '  '.join('%s %.4f' % (k, e[k]['ms_per_step']) for k in ('x0_only_u0','x0_only_u0_computed_alone','x0_only_u0_records_kept','x0_only_u0_computed_alone_records_kept')), flush=True)"
done

#!/bin/bash
# Developer tool (GPU box): the end-to-end MPC step legs of bench.py for several shapes, one line per leg
for s in "6 3 256 1024" "8 4 256 1024" "13 4 256 1024" "12 4 1024 512" "16 4 256 1024" "12 4 256 16"; do
  set -- $s
  python bench.py --no-cpu --no-configs --no-modes --nx $1 --nu $2 --horizon $3 --batch $4 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); e=d['end_to_end']
print('($1,$2,$3) x $4  solve %.4f ms |' % d['ms_per_step'], '  '.join('%s %.4f' % (k.replace('x0_only_','').replace('device_resident_','dev:').replace('_computed_alone','_alone').replace('_records_kept','+rec'), v['ms_per_step']) for k,v in e.items() if isinstance(v,dict)), flush=True)"
done

#!/bin/bash
# weight-gradient placement A/B on one box: stream-K vs the XCD-co-located table with 4..7 row ranges
export TMPDIR=/tmp
cd "$(dirname "$0")/.."
for v in streamk 0 6 7; do
  if [ $v = streamk ]; then export VITPE_WGRAD_STREAMK=1; unset VITPE_WGRAD_RANGES; else unset VITPE_WGRAD_STREAMK; export VITPE_WGRAD_RANGES=$v; fi
  echo "== $v"; timeout -k 10 200 python bench.py --steps 100 --warmup 20 --no-cpu-baseline | python3 -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); w=[o for o in d['other_kernels'] if o['name']=='wgrad_group'][0]
print(d['ms_per_step'], 'wgrad us', round(w['launch_ms']*1e3,1))"
done

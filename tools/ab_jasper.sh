#!/bin/bash
# A/B of an environment knob on ONE box: tools/ab_jasper.sh VAR A B [steps]  -> ms/step of each, twice
VAR=$1; A=$2; B=$3; STEPS=${4:-12}
for rep in 1 2; do
  for v in "$A" "$B"; do
    env $VAR=$v python bench.py --no-other-configs --no-transformer --no-cpu-baseline --steps $STEPS --warmup 4 \
      2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
r=d['roofline'].get('rest_of_step') or {}
print('$VAR=$v', 'ms/step %.2f' % d['ms_per_step'], 'frac %.3f' % d['roofline']['frac'], {k[:28]: round(x['ms_per_step'],2) for k,x in r.items() if isinstance(x,dict)})"
  done
done

#!/bin/bash
# Dev tool, run ON THE GPU BOX (round 6): bench.py (one workload, no extras) under a list of environment settings, per BASELINE config.
#   usage: r6_env_sweep.sh out.log "<wl:frames> ..." "ENV=..;ENV=.. ENV2=.." [lanes]
out=$1; wls=$2; IFS=';' read -ra ENVS <<< "$3"; lanes=${4:-0}
for wf in $wls; do
  w=${wf%%:*}; f=${wf##*:}
  for e in "${ENVS[@]}"; do
    env $e python3 bench.py --no-cpu-baseline --no-extras --workload $w --frames $f --lanes $lanes --steps 20 --warmup 10 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('$w', '[$e]', 'lanes', r.get('lanes'), 'fps', round(d['value']), 'ms', round(d['ms_per_step'],4), 'frac', round(r['frac'],4), 'p10/50/90', [round(x,3) for x in d['ms_per_step_p10_median_p90']], 'one-stream us', r.get('one_stream_us'))" >> $out || exit 1
  done
done

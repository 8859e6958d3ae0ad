#!/bin/bash
# dev tool: bench.py (headline only) under a list of environment settings;  usage: sweep_env.sh out.log "A=1" "A=2 B=3" ...
# (BENCH_ARGS adds bench.py arguments, e.g. BENCH_ARGS="--lanes 2")
out=$1; shift
for e in "$@"; do
  echo "== $e $BENCH_ARGS" >> $out
  env $e python3 bench.py --no-cpu-baseline --no-extras $BENCH_ARGS 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print(round(d['value']), round(d['ms_per_step'],4), round(r['frac'],4), [round(x,3) for x in d['ms_per_step_p10_median_p90']])" >> $out || exit 1
done

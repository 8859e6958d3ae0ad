#!/bin/bash
# dev tool, run ON THE GPU BOX: bench.py (one workload, no extras, two lanes) over frames per step for configs 3 / 4 / 5
for wf in "spot_bunny_phong_1080p:64 96 128 192 256" "spot_x16_texture_2048:32 64 96 128 192" "spot_x8_overdraw_4096:16 32 48 64 96"; do
  w=${wf%%:*}
  for f in ${wf##*:}; do
    python3 bench.py --no-cpu-baseline --no-extras --workload $w --frames $f --steps 12 --warmup 8 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$w', 'frames', $f, round(d['value']), round(d['ms_per_step'],4), round(d['roofline']['frac'],4))"
  done
done

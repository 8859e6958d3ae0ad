#!/bin/bash
# dev tool, run ON THE GPU BOX: bench.py (one workload, no extras) with 2 / 3 / 4 lanes for the four GPU configs of BASELINE.json
mkdir -p gpurun_out/r5
for wf in spot_texture_1024:256 spot_bunny_phong_1080p:128 spot_x16_texture_2048:128 spot_x8_overdraw_4096:64; do
  w=${wf%%:*}; f=${wf##*:}
  for l in 2 3 4; do
    python3 bench.py --no-cpu-baseline --no-extras --workload $w --frames $f --lanes $l --steps 20 --warmup 10 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$w', 'lanes', $l, round(d['value']), round(d['ms_per_step'],4), round(d['roofline']['frac'],4))"
  done
done

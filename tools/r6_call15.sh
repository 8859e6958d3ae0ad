#!/bin/bash
# round 6, call 15: does every set of a config pick the same grid?  (device-side measurement, SRZ_CLEAR_TRACE=1, three runs per config)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r6; mkdir -p $O; rm -f $O/call15.log $O/call15.err
for wf in readme_spot_crate_1024:256 spot_texture_1024_p7.5:256 spot_texture_1024:256 spot_bunny_phong_1080p:128 spot_x16_texture_2048:128 spot_x8_overdraw_4096:64; do
  w=${wf%%:*}; f=${wf##*:}
  for k in 1 2 3; do
    echo "== $w run $k" >> $O/call15.err
    SRZ_CLEAR_TRACE=1 python3 bench.py --no-cpu-baseline --no-extras --workload $w --frames $f --steps 20 --warmup 5 2>>$O/call15.err | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('$w', 'run $k', 'fps', round(d['value']), 'frac', round(r['frac'],4), 'unprimed', d.get('value_unprimed'), 'wgs', r.get('clear_wgs'))" >> $O/call15.log || exit 1
  done
done
grep "clear grid\|^==" $O/call15.err >> $O/call15.log
cat $O/call15.log

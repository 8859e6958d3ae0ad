#!/bin/bash
# round 6, call 9: the per-plane clear with its measured grid (SRZ_CLEAR_TRACE=1 prints the measurement) against the grid fixed at 96 and
# against the shipped interleaved clear (build/base.so), per config, two lanes; then the GPU suite's clear / frameset tests
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r6; mkdir -p $O; rm -f $O/call9.log $O/call9.err
for wf in spot_texture_1024:256 spot_bunny_phong_1080p:128 spot_x16_texture_2048:128 spot_x8_overdraw_4096:64; do
  w=${wf%%:*}; f=${wf##*:}
  for e in "SRZ_CLEAR_TRACE=1" "SRZ_CLEAR_TUNE=0" "SRZ_LIB_PATH=$PWD/software-rasterizer_amd/build/base.so" "SRZ_CLEAR_TRACE=1"; do
    env $e python3 bench.py --no-cpu-baseline --no-extras --workload $w --frames $f --steps 20 --warmup 5 2>>$O/call9.err | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('$w', '[${e##*/}]', 'lanes', r.get('lanes'), 'fps', round(d['value']), 'ms', round(d['ms_per_step'],4), 'frac', round(r['frac'],4), 'unprimed', d.get('value_unprimed'), 'one-stream us', r.get('one_stream_us'))" >> $O/call9.log || exit 1
  done
done
grep "clear grid" $O/call9.err >> $O/call9.log
cat $O/call9.log
python3 -m pytest tests/test_gpu_frameset.py tests/test_gpu_raster_paths.py tests/test_gpu_exchange.py tests/test_gpu_bench_line.py -m gpu -x -q 2>&1 | tail -5

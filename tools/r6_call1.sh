#!/bin/bash
# round 6, GPU call 1: baseline + environment-only experiments on configs 3 / 4 / 5 (two lanes, as the bench line runs them)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r6; mkdir -p $O; rm -f $O/call1.log
WL="spot_bunny_phong_1080p:128 spot_x16_texture_2048:128 spot_x8_overdraw_4096:64"
bash tools/r6_env_sweep.sh $O/call1.log "$WL" "A=0;SRZ_CLEAR_WGS=32;SRZ_CLEAR_WGS=16;SRZ_SHADE_LDS_PAD=1024;SRZ_SHADE_LDS_PAD=6144;SRZ_NO_TURNS=1;SRZ_SHADE_GRID=16384" || exit 1
echo "--- lanes 1" >> $O/call1.log
bash tools/r6_env_sweep.sh $O/call1.log "spot_x16_texture_2048:128 spot_x8_overdraw_4096:64" "A=0;SRZ_CLEAR_WGS=32;SRZ_CLEAR_WGS=16" 1 || exit 1
# timeline of the two lanes on config 4 and 5
for wf in spot_x16_texture_2048:128 spot_x8_overdraw_4096:64; do
  w=${wf%%:*}; f=${wf##*:}
  rm -rf $O/tr_$w
  rocprofv3 --kernel-trace --output-format csv -d $O/tr_$w -- python3 bench.py --no-cpu-baseline --no-extras --workload $w --frames $f --lanes 2 --steps 10 --warmup 5 > $O/tr_$w.json 2> $O/tr_$w.err || { tail -5 $O/tr_$w.err; exit 1; }
  echo "--- overlap $w" >> $O/call1.log
  python3 tools/overlap_trace.py $O/tr_$w $O/overlap_$w.json >> $O/call1.log
  rm -rf $O/tr_$w
done
cat $O/call1.log

#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r6; mkdir -p $O; rm -f $O/call3.log
WL="spot_x16_texture_2048:128 spot_x8_overdraw_4096:64 spot_bunny_phong_1080p:128"
bash tools/r6_env_sweep.sh $O/call3.log "$WL" "A=0;SRZ_CLEAR_WGS=80;SRZ_CLEAR_WGS=96;SRZ_CLEAR_WGS=128;SRZ_CLEAR_WGS=192;A=1" || exit 1
cat $O/call3.log

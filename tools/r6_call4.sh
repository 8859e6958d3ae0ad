#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r6; mkdir -p $O; rm -f $O/call4.log
WL="spot_x16_texture_2048:128 spot_x8_overdraw_4096:64 spot_texture_1024:256"
bash tools/r6_env_sweep.sh $O/call4.log "$WL" "A=0;SRZ_CLEAR_THREADS=64 SRZ_CLEAR_WGS=256;SRZ_CLEAR_THREADS=64 SRZ_CLEAR_WGS=512;SRZ_CLEAR_THREADS=64 SRZ_CLEAR_WGS=1024;SRZ_CLEAR_THREADS=128 SRZ_CLEAR_WGS=256;SRZ_CLEAR_THREADS=64 SRZ_CLEAR_WGS=384;A=1" || exit 1
cat $O/call4.log

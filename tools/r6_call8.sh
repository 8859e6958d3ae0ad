#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r6; mkdir -p $O; rm -f $O/call8.log
export SRZ_LIB_PATH=$PWD/software-rasterizer_amd/build/clrpp.so
bash tools/r6_env_sweep.sh $O/call8.log "spot_texture_1024:256 spot_bunny_phong_1080p:128 spot_x16_texture_2048:128 spot_x8_overdraw_4096:64" "SRZ_CLEAR_WGS=64;SRZ_CLEAR_WGS=80;SRZ_CLEAR_WGS=96;SRZ_CLEAR_WGS=128;SRZ_CLEAR_WGS=160;SRZ_CLEAR_WGS=192;SRZ_CLEAR_WGS=256;SRZ_CLEAR_WGS=384" || exit 1
cat $O/call8.log

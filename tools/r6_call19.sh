#!/bin/bash
# round 6, call 19: would a fourth candidate above 256 workgroups pay on the chain-bound configs?
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r6; mkdir -p $O; rm -f $O/call19.log
bash tools/r6_env_sweep.sh $O/call19.log "spot_x16_texture_2048:128" "SRZ_CLEAR_WGS=256;SRZ_CLEAR_WGS=320;SRZ_CLEAR_WGS=384;SRZ_CLEAR_WGS=512;SRZ_CLEAR_WGS=256;SRZ_CLEAR_WGS=320;SRZ_CLEAR_WGS=384;SRZ_CLEAR_WGS=512" 2 || exit 1
bash tools/r6_env_sweep.sh $O/call19.log "spot_x8_overdraw_4096:64" "SRZ_CLEAR_WGS=96;SRZ_CLEAR_WGS=128;SRZ_CLEAR_WGS=160;SRZ_CLEAR_WGS=96;SRZ_CLEAR_WGS=128;SRZ_CLEAR_WGS=160" 2 || exit 1
cat $O/call19.log

#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r6; mkdir -p $O; rm -f $O/call5.log
WL="spot_x16_texture_2048:128 spot_x8_overdraw_4096:64"
bash tools/r6_env_sweep.sh $O/call5.log "$WL" "A=0;SRZ_CLEAR_POL=1;SRZ_CLEAR_POL=1 SRZ_CLEAR_WGS=96;SRZ_CLEAR_POL=2;SRZ_CLEAR_POL=3;SRZ_CLEAR_POL=4;A=1" || exit 1
cat $O/call5.log

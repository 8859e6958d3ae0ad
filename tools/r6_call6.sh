#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r6; mkdir -p $O; rm -f $O/call6.log
bash tools/r6_env_sweep.sh $O/call6.log "spot_texture_1024:256" "A=0;SRZ_CLEAR_WGS=52;SRZ_CLEAR_WGS=56;SRZ_CLEAR_WGS=60;SRZ_CLEAR_WGS=68;SRZ_CLEAR_WGS=72;A=1;SRZ_CLEAR_WGS=60;SRZ_CLEAR_WGS=68" || exit 1
cat $O/call6.log

#!/bin/bash
# round 6, call 17: larger sub-batches (call 16: 256 frames' worth beats the shipped 192)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r6; mkdir -p $O; rm -f $O/call17.log
bash tools/r6_env_sweep.sh $O/call17.log "spot_texture_1024:512 spot_texture_1024:1024" "A=0;SRZ_SUB_BATCH=224;SRZ_SUB_BATCH=256;SRZ_SUB_BATCH=320;SRZ_SUB_BATCH=384;SRZ_SUB_BATCH=512" 1 || exit 1
bash tools/r6_env_sweep.sh $O/call17.log "spot_x16_texture_2048:256 spot_bunny_phong_1080p:512" "A=0;SRZ_SUB_BATCH=64;SRZ_SUB_BATCH=128;SRZ_SUB_BATCH=1024" 1 || exit 1
cat $O/call17.log

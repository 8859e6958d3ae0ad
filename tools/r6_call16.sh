#!/bin/bash
# round 6, call 16: the sub-batch size (frames per launch set inside one srz_frameset_render) once more, beside the per-plane clear
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r6; mkdir -p $O; rm -f $O/call16.log
bash tools/r6_env_sweep.sh $O/call16.log "spot_texture_1024:512 spot_texture_1024:1024" "A=0;SRZ_SUB_BATCH=96;SRZ_SUB_BATCH=128;SRZ_SUB_BATCH=160;SRZ_SUB_BATCH=256" 1 || exit 1
bash tools/r6_env_sweep.sh $O/call16.log "spot_texture_1024:512 spot_texture_1024:1024" "A=0;SRZ_SUB_BATCH=96;SRZ_SUB_BATCH=128;SRZ_SUB_BATCH=160" 2 || exit 1
bash tools/r6_env_sweep.sh $O/call16.log "spot_texture_1024:256" "A=0;SRZ_SUB_BATCH=64;SRZ_SUB_BATCH=96;SRZ_SUB_BATCH=128" 1 || exit 1
cat $O/call16.log

#!/bin/bash
# round 6, call 13: k_shade's grid, its residency cap and the turns once more, beside the per-plane clear (two lanes)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r6; mkdir -p $O; rm -f $O/call13.log
bash tools/r6_env_sweep.sh $O/call13.log "spot_texture_1024:256 spot_bunny_phong_1080p:128 spot_x16_texture_2048:128 spot_x8_overdraw_4096:64" "A=0;SRZ_SHADE_GRID=1024;SRZ_SHADE_GRID=1536;SRZ_SHADE_GRID=3072;SRZ_SHADE_GRID=4096;SRZ_SHADE_LDS_PAD=1024;SRZ_NO_TURNS=1;A=1" 2 || exit 1
cat $O/call13.log

#!/bin/bash
# round 6, call 14: config 2, k_shade's grid / residency cap beside the per-plane clear, repeated (is call 13's +3 % real?)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r6; mkdir -p $O; rm -f $O/call14.log
E1="A=0;SRZ_SHADE_GRID=1024;SRZ_SHADE_LDS_PAD=1024;SRZ_SHADE_GRID=4096;SRZ_SHADE_GRID=768;SRZ_SHADE_GRID=1280;SRZ_SHADE_LDS_PAD=6144"
bash tools/r6_env_sweep.sh $O/call14.log "spot_texture_1024:256" "$E1;$E1;SRZ_SHADE_GRID=1024 SRZ_CLEAR_WGS=112;SRZ_SHADE_GRID=1024 SRZ_CLEAR_WGS=96;SRZ_SHADE_GRID=1024 SRZ_CLEAR_WGS=80" 2 || exit 1
bash tools/r6_env_sweep.sh $O/call14.log "readme_spot_crate_1024:256 spot_bump_1024:256" "A=0;SRZ_SHADE_GRID=1024;SRZ_SHADE_LDS_PAD=1024" 2 || exit 1
cat $O/call14.log

#!/bin/bash
# round 6, call 12: lanes and frames per step once more, with the per-plane clear (measured grid)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r6; mkdir -p $O; rm -f $O/call12.log
for l in 2 3 1; do
  bash tools/r6_env_sweep.sh $O/call12.log "spot_texture_1024:256 spot_bunny_phong_1080p:128 spot_x16_texture_2048:128 spot_x8_overdraw_4096:64" "L=$l" $l || exit 1
done
bash tools/r6_env_sweep.sh $O/call12.log "spot_texture_1024:128 spot_texture_1024:192 spot_texture_1024:384 spot_texture_1024:512 spot_x16_texture_2048:64 spot_x16_texture_2048:96 spot_x8_overdraw_4096:32 spot_x8_overdraw_4096:96" "L=2" 2 || exit 1
cat $O/call12.log

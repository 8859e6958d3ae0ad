#!/bin/bash
# round 6, call 20: the per-plane clear's workgroup size (SRZ_CLEAR_THREADS) at equal numbers of waves
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r6; mkdir -p $O; rm -f $O/call20.log
bash tools/r6_env_sweep.sh $O/call20.log "spot_texture_1024:256" "SRZ_CLEAR_WGS=96;SRZ_CLEAR_THREADS=128 SRZ_CLEAR_WGS=192;SRZ_CLEAR_THREADS=128 SRZ_CLEAR_WGS=160;SRZ_CLEAR_THREADS=64 SRZ_CLEAR_WGS=384;SRZ_CLEAR_THREADS=64 SRZ_CLEAR_WGS=320;SRZ_CLEAR_THREADS=192 SRZ_CLEAR_WGS=128;SRZ_CLEAR_WGS=96" 2 || exit 1
bash tools/r6_env_sweep.sh $O/call20.log "spot_x16_texture_2048:128" "SRZ_CLEAR_WGS=256;SRZ_CLEAR_THREADS=128 SRZ_CLEAR_WGS=512;SRZ_CLEAR_THREADS=128 SRZ_CLEAR_WGS=384;SRZ_CLEAR_THREADS=64 SRZ_CLEAR_WGS=1024;SRZ_CLEAR_WGS=256" 2 || exit 1
cat $O/call20.log

#!/bin/bash
# round 6, call 18: the shipped sub-batch size (256 frames' worth) on large single sets, and the sub-batch test
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r6; mkdir -p $O; rm -f $O/call18.log
bash tools/r6_env_sweep.sh $O/call18.log "spot_texture_1024:320 spot_texture_1024:384 spot_texture_1024:512 spot_texture_1024:1024 spot_bunny_phong_1080p:512" "A=0;SRZ_SUB_BATCH=192" 1 || exit 1
cat $O/call18.log
python3 -m pytest tests/test_gpu_frameset.py -m gpu -x -q 2>&1 | tail -2

#!/bin/bash
# round 6, call 10: order of the clear's work items — planes of a band next to each other (shipped) / a frame's planes one after the other (-DSRZ_CLEAR_PLANE_MAJOR, gone)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r6; mkdir -p $O; rm -f $O/call10.log
B=$PWD/software-rasterizer_amd/build
run() { # workload frames wgs
  for l in cur clrpm cur clrpm; do
    SRZ_LIB_PATH=$B/$l.so bash tools/r6_env_sweep.sh $O/call10.log "$1:$2" "SRZ_CLEAR_WGS=$3 L=$l" || exit 1
  done
}
run spot_bunny_phong_1080p 128 96
run spot_x8_overdraw_4096 64 160
cat $O/call10.log

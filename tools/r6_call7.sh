#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r6; mkdir -p $O; rm -f $O/call7.log
for lib in - clrpp.so - clrpp.so; do
  if [ "$lib" = "-" ]; then unset SRZ_LIB_PATH; else export SRZ_LIB_PATH=$PWD/software-rasterizer_amd/build/$lib; fi
  echo "== $lib" >> $O/call7.log
  bash tools/r6_env_sweep.sh $O/call7.log "spot_texture_1024:256 spot_x16_texture_2048:128" "A=0;SRZ_CLEAR_WGS=96" || exit 1
done
cat $O/call7.log

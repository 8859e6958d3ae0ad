#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
export SRZ_LIB_PATH=$PWD/software-rasterizer_amd/build/probe.so
for c in "4 128" "5 64" "2 256"; do
  echo "== beside the clear (64 workgroups)"; python3 tools/phase_probe.py $c 2>&1 | tail -6
  echo "== clear throttled to 2 workgroups (k_shade nearly alone)"; SRZ_CLEAR_WGS=2 python3 tools/phase_probe.py $c 2>&1 | tail -6
done

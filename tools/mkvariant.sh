#!/bin/bash
# Dev tool (CPU): build a variant of libsrz.so for a same-box A/B (tools/ab.sh, tools/lib_sweep.sh): mkvariant.sh <name> [-DFLAG ...]
cd "$(dirname "$0")/../software-rasterizer_amd" && mkdir -p build
n=$1; shift
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -Wall -Wno-unused-parameter "$@" -shared -o build/$n.so csrc/srz_kernels.hip csrc/srz_api.hip -ldl

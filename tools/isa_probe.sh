#!/bin/bash
# Dev tool: static instruction mix of the two per-pixel shading paths (probe kernels under SRZ_ISA_PROBE), i.e. the
# dynamic VALU count per 64-pixel chunk for the TEXTURE shader with 2 lights and p = 150.
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p $R/gpurun_out/isa
SH=${1:--1}   # (second argument `approx`: the tolerance mode's ApproxMath instead of FastMath; third argument `grey`: FrameK::grey set)
# -1: generic per-pixel build; 0 NORMAL / 1 TEXTURE / 2 PHONG: the FAST variant (2 lights, p = 150)
for v in V S; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -DSRZ_ISA_PROBE -DSRZ_PROBE_SH=$SH \
    $([ $v = S ] && echo -DSRZ_PROBE_S) $([ "$2" = approx ] && echo -DSRZ_PROBE_APPROX) $([ "$3" = grey ] && echo -DSRZ_PROBE_GREY) -I$R/include --cuda-device-only -S $R/software-rasterizer_amd/csrc/srz_kernels.hip \
    -o $R/gpurun_out/isa/probe_$v.s
done
python3 - <<PY
import collections
print("shader variant", $SH)
for v in "VS":
    s=open('$R/gpurun_out/isa/probe_%s.s' % v).read()
    i=s.index('_ZN3srz7probe_vENS_10RenderArgsEPf:'); j=s.index('.Lfunc_end',i)
    lines=[l.strip() for l in s[i:j].split('\n')]
    ins=[l.split()[0] for l in lines if l and not l.startswith(('.',';','_')) and not l.endswith(':')]
    c=collections.Counter(ins)
    valu=sum(n for k,n in c.items() if k.startswith('v_'))
    trans=sum(n for k,n in c.items() if k.startswith(('v_rcp','v_sqrt','v_rsq','v_exp','v_log')))
    f64=sum(n for k,n in c.items() if 'f64' in k)
    print(v,'total',len(ins),'valu',valu,'trans',trans,'f64',f64,'salu',sum(n for k,n in c.items() if k.startswith('s_')), 'branches', sum(n for k,n in c.items() if 'branch' in k))
    print(' ', c.most_common(28))
PY

#!/bin/bash
# dev tool, run ON THE GPU BOX: bench.py (headline workload only) with a list of library builds
#   usage: lib_sweep.sh out.log "<bench args>" libA.so libB.so ...   ("-" = the shipped libsrz.so; names relative to software-rasterizer_amd/build/)
out=$1; args=$2; shift 2
for lib in "$@"; do
  if [ "$lib" = "-" ]; then unset SRZ_LIB_PATH; else export SRZ_LIB_PATH=$PWD/software-rasterizer_amd/build/$lib; fi
  echo "== $lib $args" >> $out
  python3 bench.py --no-cpu-baseline --no-extras $args 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print(round(d['value']), round(d['ms_per_step'],4), round(r['frac'],4), [round(x,3) for x in d['ms_per_step_p10_median_p90']], 'one-stream us [setup+bin, raster, shade]', r['one_stream_us'])" >> $out || exit 1
done

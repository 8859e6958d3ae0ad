#!/bin/bash
# dev tool: bench.py (headline only) with each of a list of library builds;  usage: sweep_libs.sh out.log "libA.so libB.so -" [bench args]
out=$1; libs=$2; shift; shift
for l in $libs; do
  if [ "$l" = "-" ]; then unset SRZ_LIB_PATH; else export SRZ_LIB_PATH=$PWD/software-rasterizer_amd/build/$l; fi
  echo "== $l $*" >> $out
  python3 bench.py --no-cpu-baseline --no-extras "$@" 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; o=r['one_stream']
print(round(d['value']), round(d['ms_per_step'],4), round(r['frac'],4), [round(x,3) for x in d['ms_per_step_p10_median_p90']], 'one-stream', round(o['ms_per_step'],4), 'bin/raster/shade', round(o['k_setup_bin_ms'],3), round(o['k_raster_ms'],3), round(o['k_shade_ms'],3))" >> $out || exit 1
done

#!/bin/bash
# Dev tool, run ON THE GPU BOX: same-box A/B of library builds.  usage: bash tools/ab.sh "libA.so libB.so" "2 256;4 32"
# (lib names relative to software-rasterizer_amd/build/, "-" = the shipped libsrz.so); prints per-kernel average µs per render
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
IFS=';' read -ra CASES <<< "$2"
for c in "${CASES[@]}"; do
  for lib in $1; do
    if [ "$lib" = "-" ]; then unset SRZ_LIB_PATH; else export SRZ_LIB_PATH=$PWD/software-rasterizer_amd/build/$lib; fi
    O=gpurun_out/ab_$(echo "$lib$c" | tr -c 'a-zA-Z0-9' '_')
    rm -rf $O; mkdir -p $O
    rocprofv3 --kernel-trace --stats --output-format csv -d $O/t -- python3 tools/perf_probe.py $c 10 > $O/run.log 2>&1 || { tail -5 $O/run.log; exit 1; }
    python3 - "$O" "$lib" "$c" <<'PY'
import csv, glob, re, sys
f = glob.glob(sys.argv[1] + "/t/**/*kernel_stats.csv", recursive=True)[0]
t = {}
for r in csv.DictReader(open(f)):
    n = r["Name"]
    if "<true" in n or "rocclr" in n: continue
    k = re.sub(r"^void |srz::|\(.*$", "", n)
    t[k] = float(r["AverageNs"]) / 1e3
log = open(sys.argv[1] + "/run.log").read()
m = re.search(r"wall/render=([\d.]+) ms.*total_ms=([\d.]+)", log)
keys = ["k_setup<false>", "k_chunks", "k_bin", "k_raster<1>", "k_raster_slow<false>", "k_shade<false, 2, false, false>", "k_shade<false, 0, false, false>", "k_clear"]
shade_fast = sum(v for k, v in t.items() if k.startswith("k_shade<false,") and not k.startswith("k_shade<false, 0,"))
print(f"{sys.argv[3]:8s} {sys.argv[2]:18s} " + " ".join(f"{k.split('<')[0][2:]}{'F' if 'true>' in k else ''}={t.get(k, 0):7.1f}" for k in keys if k in t and "k_shade" not in k) + f" shade_fast={shade_fast:7.1f}" + (f"  wall={float(m.group(1))*1e3:7.1f} events={float(m.group(2))*1e3:7.1f}" if m else ""))
PY
  done
done

#!/bin/bash
# Dev tool, run ON THE GPU BOX: same-box A/B of environment settings with the shipped library.
# usage: bash tools/ab_env.sh "SRZ_FUSE=0;SRZ_FUSE=1" "2 256;4 32"   (prints per-kernel average µs per render, one stream)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
IFS=';' read -ra CASES <<< "$2"
IFS=';' read -ra ENVS <<< "$1"
for c in "${CASES[@]}"; do
  for e in "${ENVS[@]}"; do
    O=gpurun_out/abe_$(echo "$e$c" | tr -c 'a-zA-Z0-9' '_')
    rm -rf $O; mkdir -p $O
    env $e rocprofv3 --kernel-trace --stats --output-format csv -d $O/t -- python3 tools/perf_probe.py $c 10 > $O/run.log 2>&1 || { tail -5 $O/run.log; exit 1; }
    python3 - "$O" "$e" "$c" <<'PY'
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
print(f"{sys.argv[3]:8s} {sys.argv[2]:22s} " + " ".join(f"{k[2:]}={v:7.1f}" for k, v in sorted(t.items()) if k.startswith("k_") and "tex_convert" not in k) + (f"  wall={float(m.group(1))*1e3:7.1f} events={float(m.group(2))*1e3:7.1f}" if m else ""))
PY
  done
done

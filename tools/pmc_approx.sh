#!/bin/bash
# dev tool, run ON THE GPU BOX: dynamic VALU instructions / wave cycles of k_shade, exact against tolerance mode (usage: pmc_approx.sh "5 64" "4 128")
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for c in "$@"; do
  for m in exact approx; do
    O=gpurun_out/pmca_$(echo "$c$m" | tr -c 'a-zA-Z0-9' '_'); rm -rf $O; mkdir -p $O
    if [ $m = approx ]; then export PROBE_APPROX=1; else unset PROBE_APPROX; fi
    rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_SALU --output-format csv -d $O/p -- python3 tools/perf_probe.py $c 6 > $O/run.log 2>&1 || { tail -5 $O/run.log; exit 1; }
    python3 - "$O" "$c" "$m" <<'PY'
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/p/**/*counter_collection.csv", recursive=True)[0]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    n = r["Kernel_Name"]
    if "k_shade" not in n or "<true" in n: continue
    agg[n.split("(")[0].replace("void srz::", "")][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in agg.items():
    v = {c: sum(x[len(x)//4:]) / max(1, len(x) - len(x)//4) for c, x in d.items()}
    if v.get("SQ_INSTS_VALU", 0) > 1e6:
        print(f"{sys.argv[2]:8s} {sys.argv[3]:6s} {k:36s} VALU {v['SQ_INSTS_VALU']/1e6:8.1f} M  SALU {v['SQ_INSTS_SALU']/1e6:7.1f} M  wave-cycles {v['SQ_WAVE_CYCLES']/1e6:8.1f} M  waiting {v['SQ_WAIT_ANY']/v['SQ_WAVE_CYCLES']:.2f}")
PY
  done
done

#!/bin/bash
# Dev tool, run ON THE GPU BOX: per-kernel durations of tools/perf_probe.py (usage: bash tools/trace_probe.sh "2 256" tag)
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/trace_$2
rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/t -- python3 tools/perf_probe.py $1 10 > $O/run.log 2>&1 || { tail -5 $O/run.log; exit 1; }
tail -2 $O/run.log
python3 - "$O" <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/t/**/*kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    if "<true>" in r["Name"] or "rocclr" in r["Name"]: continue
    print(f'{r["Name"][:60]:60s} calls {r["Calls"]:>4s} avg {float(r["AverageNs"])/1e3:9.1f} us  min {float(r["MinNs"])/1e3:9.1f}')
PY

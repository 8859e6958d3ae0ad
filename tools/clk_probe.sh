#!/bin/bash
# Dev tool, run ON THE GPU BOX: effective shader clock per kernel = GRBM_GUI_ACTIVE / 8 XCDs / kernel duration
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/clk_$2
rm -rf $O; mkdir -p $O
rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/g -- python3 tools/perf_probe.py $1 6 > $O/run.log 2>&1 || { tail -5 $O/run.log; exit 1; }
python3 - "$O" <<'PY'
import csv, glob, sys, collections
d = sys.argv[1]
cc = glob.glob(d + "/g/**/*counter_collection.csv", recursive=True)[0]
kt = glob.glob(d + "/g/**/*kernel_trace.csv", recursive=True)[0]
dur = {}
for r in csv.DictReader(open(kt)):
    dur[r["Dispatch_Id"]] = (r["Kernel_Name"], int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
agg = collections.defaultdict(list)
for r in csv.DictReader(open(cc)):
    if r["Counter_Name"] != "GRBM_GUI_ACTIVE": continue
    n, ns = dur.get(r["Dispatch_Id"], (None, 0))
    if n and ns > 20000 and "<true" not in n:
        agg[n[:50]].append(float(r["Counter_Value"]) / 8.0 / ns)
for n, v in agg.items():
    print(f"{n:50s} {sum(v)/len(v):.3f} GHz effective ({len(v)} dispatches)")
PY

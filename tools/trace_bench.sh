# Dev tool, run ON THE GPU BOX: kernel durations of every render of a default bench.py run (main case + extras), in order
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/tb; rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tb -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/tb.json 2> gpurun_out/tb.err
python3 - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/tb/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# print per-render sequences compactly: for each k_setup start a new line
line = []
out = []
for r in rows:
    n = r["Kernel_Name"]
    short = None
    for k in ("k_setup", "k_bin", "k_raster_slow", "k_raster", "k_clear", "k_shade", "k_vertex"):
        if k in n:
            short = k; break
    if not short: continue
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    if short == "k_setup" and line:
        out.append(line); line = []
    line.append(f"{short[2:6]}={d:.0f}")
out.append(line)
import collections
prev = None
for i, l in enumerate(out):
    s = " ".join(l)
    if i % 3 == 0: print(i, s)
PY

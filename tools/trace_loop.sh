# Dev tool, run ON THE GPU BOX: the kernel timeline of single-frame draws through the C++ API (loop_bench, draw + finish per frame)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/tl; rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d gpurun_out/tl -- software-rasterizer_amd/build/loop_bench . 60 readme > gpurun_out/tl.json 2> gpurun_out/tl.err
python3 - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/tl/**/*kernel_trace.csv", recursive=True)[0]
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][-40:]) for r in csv.DictReader(open(f))]
for g in glob.glob("gpurun_out/tl/**/*memory_copy_trace.csv", recursive=True):
    rows += [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "COPY " + r.get("Direction", "")) for r in csv.DictReader(open(g))]
rows.sort()
# the frames of the draw_complete pass (one frame in flight): print three frames from the middle of the run
starts = [i for i, r in enumerate(rows) if "k_vertex" in r[2]]
for fi in starts[200:203]:
    t0 = rows[fi][0]
    j = fi - 1 if fi > 0 and "COPY" in rows[fi - 1][2] else fi
    t0 = rows[j][0]
    print("frame:")
    while j < len(rows) and (j <= fi or "k_vertex" not in rows[j][2]):
        s, e, n = rows[j]
        print(f"   +{(s - t0) / 1e3:7.1f} us  {(e - s) / 1e3:6.1f} us  {n}")
        j += 1
PY

#!/usr/bin/env python3
"""Turn gpurun_out/prof_<tag>/ (written by profiles/collect.sh on the GPU box) into the committed summaries:
profiles/<tag>_kernel_stats.csv, profiles/<tag>_pmc.json, and profiles/pmc_traffic.json (read by bench.py).

HBM traffic follows /opt/skills/guides/MI355X_MICROARCH.md §HBM: FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 WRITE_SIZE
is byte-exact for 16-byte-per-lane streaming stores (our framebuffer stores), FETCH_SIZE reports 1/2 of the bytes of wide
coalesced reads, so the read side is given both raw and doubled; other widths are uncalibrated (stated in the json)."""
import collections
import csv
import glob
import json
import os
import shutil
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
here = os.path.dirname(os.path.abspath(__file__))
src = os.path.join(os.path.dirname(here), "gpurun_out", f"prof_{tag}")
KERNELS = ("k_setup", "k_bands", "k_raster", "k_clear", "k_shade")


def one(pattern):
    g = glob.glob(os.path.join(src, pattern), recursive=True)
    assert g, pattern
    return max(g, key=os.path.getmtime)  # (gpurun merges runs into the same scratch directory: take the latest)


shutil.copy(one("trace/**/*kernel_stats.csv"), os.path.join(here, f"{tag}_kernel_stats.csv"))
stats = {r["Name"]: r for r in csv.DictReader(open(one("trace/**/*kernel_stats.csv")))}
per_kernel_us = {}
for name, r in stats.items():
    for k in KERNELS:
        if k in name and "<true>" not in name:
            per_kernel_us[k] = float(r["AverageNs"]) / 1e3


def pmc(dirname):
    rows = list(csv.DictReader(open(one(f"{dirname}/**/*counter_collection.csv"))))
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in rows:
        for k in KERNELS:
            if k in r["Kernel_Name"] and "<true>" not in r["Kernel_Name"]:
                agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return {k: {c: sum(v) / len(v) for c, v in d.items()} for k, d in agg.items()}


fetch, write, sq = pmc("pmc_fetch"), pmc("pmc_write"), pmc("pmc_sq")
bench = json.loads(open(os.path.join(src, "bench_plain.json")).read().strip().splitlines()[-1])
out = {"tag": tag, "bench_line": bench, "avg_kernel_us": per_kernel_us, "pipeline_us_sum": sum(v for k, v in per_kernel_us.items() if k != "k_clear"),
       "note": "k_clear runs on a second stream beside k_raster/k_shade: its time overlaps theirs and is not in pipeline_us_sum",
       "pmc_avg_per_launch": {}, "units": "FETCH_SIZE/WRITE_SIZE in KiB (rocprofv3); bytes below = KiB*1024"}
tot_w = tot_f = 0.0
for k in KERNELS:
    f = fetch.get(k, {}).get("FETCH_SIZE", 0.0) * 1024
    w = write.get(k, {}).get("WRITE_SIZE", 0.0) * 1024
    tot_f += f
    tot_w += w
    out["pmc_avg_per_launch"][k] = {"fetch_bytes_raw": f, "write_bytes": w, **sq.get(k, {})}
algo = bench["roofline"]["algorithmic_bytes_per_launch"]
out["hbm_bytes_per_launch"] = {"write": tot_w, "fetch_raw": tot_f, "fetch_x2_gfx950_correction": 2 * tot_f,
                               "total_corrected": tot_w + 2 * tot_f, "algorithmic": algo,
                               "ratio_traffic_over_algorithmic": (tot_w + 2 * tot_f) / algo}
json.dump(out, open(os.path.join(here, f"{tag}_pmc.json"), "w"), indent=1)
tfile = os.path.join(here, "pmc_traffic.json")
t = json.load(open(tfile)) if os.path.exists(tfile) else {}
t[bench["config"]["workload"]] = {"hbm_bytes_per_launch": tot_w + 2 * tot_f, "from": f"{tag}_pmc.json",
                                  "frames_per_launch": bench["config"]["frames_per_step"]}
json.dump(t, open(tfile, "w"), indent=1)
print(json.dumps(out["hbm_bytes_per_launch"], indent=1))
print(per_kernel_us, "events total ms:", bench["roofline"]["launch_ms"])

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
# (matched in this order: "k_raster_slow" before "k_raster"; k_shade_fast / k_shade_generic = the two builds of k_shade)
KERNELS = ("k_setup", "k_bin", "k_raster_slow", "k_raster", "k_clear", "k_shade_fast", "k_shade_generic")


def kernel_of(name):
    if "<true" in name:  # counting variants (run once, outside the timed region)
        return None
    if "k_shade" in name:
        return "k_shade_fast" if "k_shade<false, true>" in name else "k_shade_generic"
    for k in KERNELS:
        if k in name:
            return k
    return None


def one(pattern):
    g = glob.glob(os.path.join(src, pattern), recursive=True)
    assert g, pattern
    return max(g, key=os.path.getmtime)  # (gpurun merges runs into the same scratch directory: take the latest)


shutil.copy(one("trace/**/*kernel_stats.csv"), os.path.join(here, f"{tag}_kernel_stats.csv"))
try:  # the default command (two lanes on two streams: the kernels of the lanes overlap; half a batch per launch)
    shutil.copy(one("trace_lanes/**/*kernel_stats.csv"), os.path.join(here, f"{tag}_lanes_kernel_stats.csv"))
except AssertionError:
    pass
stats = {r["Name"]: r for r in csv.DictReader(open(one("trace/**/*kernel_stats.csv")))}
per_kernel_us = {}
for name, r in stats.items():
    k = kernel_of(name)
    if k:
        per_kernel_us[k] = float(r["AverageNs"]) / 1e3


def pmc(dirname):
    rows = list(csv.DictReader(open(one(f"{dirname}/**/*counter_collection.csv"))))
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in rows:
        k = kernel_of(r["Kernel_Name"])
        if k:
            agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return {k: {c: sum(v) / len(v) for c, v in d.items()} for k, d in agg.items()}


fetch, write, sq = pmc("pmc_fetch"), pmc("pmc_write"), pmc("pmc_sq")
bench = json.loads(open(os.path.join(src, "bench_plain.json")).read().strip().splitlines()[-1])
traced = json.loads(open(os.path.join(src, "bench_traced.json")).read().strip().splitlines()[-1])  # the --lanes 1 run under the tracer
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
print(per_kernel_us, "one-stream events total ms:", bench["roofline"]["one_stream"]["launch_ms"], "| traced --lanes 1 run:", traced["roofline"]["launch_ms"])

# ---- the other BASELINE configs (tests/perf_probe.py under the same two profiler modes) ---------------------------------
others = {}
for n, label in ((3, "spot_bunny_phong_1080p x64"), (4, "spot_x16_texture_2048 x32"), (5, "spot_x8_overdraw_4096 x16")):
    try:
        shutil.copy(one(f"trace_c{n}/**/*kernel_stats.csv"), os.path.join(here, f"{tag}_c{n}_kernel_stats.csv"))
        us = {}
        for r in csv.DictReader(open(one(f"trace_c{n}/**/*kernel_stats.csv"))):
            k = kernel_of(r["Name"])
            if k:
                us[k] = float(r["AverageNs"]) / 1e3
        others[f"config{n}"] = {"workload": label, "avg_kernel_us": us, "sq_avg_per_launch": pmc(f"pmc_sq_c{n}")}
    except AssertionError:
        pass
if others:
    json.dump(others, open(os.path.join(here, f"{tag}_configs_pmc.json"), "w"), indent=1)
    for k, v in others.items():
        print(k, {a: round(b) for a, b in v["avg_kernel_us"].items()})

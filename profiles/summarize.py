#!/usr/bin/env python3
"""Turn gpurun_out/prof_<tag>/ (written by profiles/collect.sh on the GPU box) into the committed summaries:

  profiles/<tag>_kernel_stats.csv          rocprofv3 --kernel-trace --stats of the headline command, one stream (--lanes 1)
  profiles/<tag>_lanes_kernel_stats.csv    the same for the default two-lane command
  profiles/<tag>_<workload>_kernel_stats.csv   for every configs[] workload
  profiles/<tag>_pmc.json                  per workload and kernel: average duration, FETCH_SIZE / WRITE_SIZE bytes, SQ counters
  profiles/pmc_counters.json               per workload: HBM bytes and VALU instructions per step (read by bench.py → roofline.traffic
                                           and valu_frac of the headline and of every configs[] entry)

HBM traffic follows /opt/skills/guides/MI355X_MICROARCH.md §HBM: FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 WRITE_SIZE is
byte-exact for 16-byte-per-lane streaming stores (the framebuffer stores), FETCH_SIZE reports 1/2 of the bytes of wide coalesced
reads, so the read side is doubled; other widths are uncalibrated (stated in the json)."""
import collections
import csv
import glob
import json
import os
import shutil
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r03"
here = os.path.dirname(os.path.abspath(__file__))
src = os.path.join(os.path.dirname(here), "gpurun_out", f"prof_{tag}")
KERNELS = ("k_setup", "k_chunks", "k_vertex", "k_bin", "k_raster_slow", "k_raster", "k_clear", "k_shade_fast", "k_shade_generic")


def kernel_of(name):
    if "<true" in name:  # counting variants (run once, outside the timed region)
        return None
    if "k_clear_tune" in name:  # (the one-thread kernel that ends a render while a set measures its clear's grid: not a step's work)
        return None
    if "k_shade" in name:
        return "k_shade_generic" if "k_shade<false, 0," in name else "k_shade_fast"  # (FAST builds: <false, lights 1..4, bumpy>)
    for k in KERNELS:
        if k in name:
            return k
    return None


def one(pattern):
    g = glob.glob(os.path.join(src, pattern), recursive=True)
    assert g, pattern
    return max(g, key=os.path.getmtime)  # (gpurun merges runs into the same scratch directory: take the latest)


def kernel_us(path):
    us = {}
    for r in csv.DictReader(open(path)):
        k = kernel_of(r["Name"])
        if k:
            us[k] = us.get(k, 0.0) + float(r["AverageNs"]) / 1e3
    return us


def pmc(dirname):
    """per kernel: counter → average per dispatch (the steady-state dispatches: the first quarter of them is warm-up)"""
    rows = list(csv.DictReader(open(one(f"{dirname}/**/*counter_collection.csv"))))
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in rows:
        k = kernel_of(r["Kernel_Name"])
        if k:
            agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return {k: {c: sum(v[len(v) // 4:]) / max(1, len(v) - len(v) // 4) for c, v in d.items()} for k, d in agg.items()}


bench = json.loads(open(os.path.join(src, "bench_plain.json")).read().strip().splitlines()[-1])
head = bench["config"]["workload"]
out = {"tag": tag, "bench_line": bench, "units": "FETCH_SIZE / WRITE_SIZE in KiB (rocprofv3); bytes below = KiB * 1024; "
       "durations: µs per dispatch = per step (one frameset on one stream: bench.py --lanes 1); k_clear runs on a second stream beside "
       "k_raster / k_shade and is not part of pipeline_us_sum", "workloads": {}}
counters_file = os.path.join(here, "pmc_counters.json")
counters = json.load(open(counters_file)) if os.path.exists(counters_file) else {}
try:
    shutil.copy(one("trace_lanes/**/*kernel_stats.csv"), os.path.join(here, f"{tag}_lanes_kernel_stats.csv"))
except AssertionError:
    pass
for jf in sorted(glob.glob(os.path.join(src, "*.traced.json"))):
    w = os.path.basename(jf)[:-len(".traced.json")]
    try:
        line = json.loads(open(jf).read().strip().splitlines()[-1])
        stats = one(f"{w}/trace/**/*kernel_stats.csv")
        shutil.copy(stats, os.path.join(here, f"{tag}_kernel_stats.csv" if w == head else f"{tag}_{w}_kernel_stats.csv"))
        us = kernel_us(stats)
        fetch, write, sq = pmc(f"{w}/pmc_fetch"), pmc(f"{w}/pmc_write"), pmc(f"{w}/pmc_sq")
        try:
            mix, occ = pmc(f"{w}/pmc_mix"), pmc(f"{w}/pmc_occ")
        except AssertionError:
            mix, occ = {}, {}
    except (AssertionError, ValueError, IndexError) as e:
        print("skipped", w, e)
        continue
    per_kernel, tot_f, tot_w, tot_valu, tot_pipe = {}, 0.0, 0.0, 0.0, 0.0
    for k in KERNELS:
        if k not in us and k not in sq:
            continue
        f = fetch.get(k, {}).get("FETCH_SIZE", 0.0) * 1024
        wr = write.get(k, {}).get("WRITE_SIZE", 0.0) * 1024
        tot_f, tot_w, tot_valu = tot_f + f, tot_w + wr, tot_valu + sq.get(k, {}).get("SQ_INSTS_VALU", 0.0)
        # VALU pipe cycles, a lower bound: 2 per wave-instruction, 2 more for binary64 arithmetic, 6 more for a transcendental
        # (tools/cpp/valu_rate_probe.hip; packed f32 instructions also take 4 but have no counter of their own)
        mk = mix.get(k, {})
        f64 = sum(mk.get(c, 0.0) for c in ("SQ_INSTS_VALU_ADD_F64", "SQ_INSTS_VALU_MUL_F64", "SQ_INSTS_VALU_FMA_F64"))
        pipe = 2.0 * sq.get(k, {}).get("SQ_INSTS_VALU", 0.0) + 2.0 * f64 + 6.0 * (mk.get("SQ_INSTS_VALU_TRANS_F32", 0.0) + mk.get("SQ_INSTS_VALU_TRANS_F64", 0.0))
        tot_pipe += pipe if mk else 0.0
        per_kernel[k] = {"avg_us": us.get(k), "fetch_bytes_raw": f, "write_bytes": wr, **sq.get(k, {}), **mk,
                         "valu_pipe_cycles_lower_bound": pipe if mk else None,
                         "valu_pipe_frac_of_kernel_time": (pipe / (1024 * 2.4e9 * us[k] * 1e-6)) if mk and us.get(k) else None,
                         "resident_waves_per_cu": occ.get(k, {}).get("MeanOccupancyPerCU")}  # (of 4 SIMDs; k_clear: 64 workgroups of 4 waves = 1.0)
    algo = line["roofline"]["algorithmic_bytes_per_launch"]
    frames = line["config"]["frames_per_step"]
    out["workloads"][w] = {"frames_per_step": frames, "kernels": per_kernel,
                           "pipeline_us_sum": sum(v for k, v in us.items() if k != "k_clear"),
                           "traced_run": {"ms_per_step": line["ms_per_step"], "launch_ms": line["roofline"]["launch_ms"]},
                           "hbm_bytes_per_step": {"write": tot_w, "fetch_raw": tot_f, "fetch_x2_gfx950_correction": 2 * tot_f,
                                                  "total_corrected": tot_w + 2 * tot_f, "algorithmic": algo,
                                                  "ratio_traffic_over_algorithmic": (tot_w + 2 * tot_f) / algo},
                           "valu_wave_insts_per_step": tot_valu,
                           "valu_per_64_visible_px": tot_valu / max(1.0, line["visible_pixels_per_frame"] * frames / 64.0)}
    counters[w] = {"frames_per_step": frames, "hbm_bytes_per_step": tot_w + 2 * tot_f, "valu_wave_insts_per_step": tot_valu,
                   "valu_pipe_cycles_per_step": tot_pipe or None,
                   "from": f"{tag}_pmc.json"}
    print(f"{w:28s} pipeline {out['workloads'][w]['pipeline_us_sum']:8.1f} us  traffic/algorithmic {(tot_w + 2 * tot_f) / algo:.3f}  "
          f"VALU {tot_valu / 1e6:7.1f} M  per 64 visible px {out['workloads'][w]['valu_per_64_visible_px']:.0f}  "
          + " ".join(f"{k[2:]}={v:.0f}" for k, v in us.items()))
# the counters describe THESE kernel sources: bench.py reports them only for a tree whose sources hash the same
sys.path.insert(0, os.path.dirname(here))
import bench  # noqa: E402
# (the hash the PROFILED run reported on the box; an old collection without one: this tree's library)
try:
    _line = [l for l in open(os.path.join(src, "bench_plain.json")) if l.startswith('{"metric"')][-1]
    _h = json.loads(_line)["roofline"].get("kernel_source_hash")
except (OSError, IndexError, KeyError, ValueError):
    _h = None
counters["_kernel_source_hash"] = out["kernel_source_hash"] = _h or bench.kernel_source_hash()
json.dump(out, open(os.path.join(here, f"{tag}_pmc.json"), "w"), indent=1)
json.dump(counters, open(counters_file, "w"), indent=1)
shutil.copy(os.path.join(src, "bench_plain.json"), os.path.join(here, f"{tag}_bench.json"))
for extra in ("bench_plain_details.json", "bench_driver_args.json", "bench_driver_args_details.json"):  # (the full records beside the compact lines)
    if os.path.exists(os.path.join(src, extra)):
        shutil.copy(os.path.join(src, extra), os.path.join(here, f"{tag}_{extra}"))

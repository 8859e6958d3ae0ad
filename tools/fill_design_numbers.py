"""dev tool: (re)write the numbers tables of DESIGN.md §5 between the markers <!-- numbers:begin --> and <!-- numbers:end -->:
  * "driver": what the DRIVER's own run of bench.py recorded — the newest BENCH_rNN.json at the repo root (written by the driver at the
    end of round NN, i.e. for the code of THAT round; none exists yet for the round in progress): its parsed line's scalars;
  * "builder": the FULL record of the builder's own run of the same command (bench_details.json; default: the newest
    profiles/rNN_bench_driver_args_details.json), per workload.
usage: python tools/fill_design_numbers.py [builder details json]"""
import json
import os
import re
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
import glob
src = sys.argv[1] if len(sys.argv) > 1 else sorted(glob.glob(os.path.join(REPO, "profiles", "r??_bench_driver_args_details.json")))[-1]
tag = os.path.basename(src)[:3]
d = json.load(open(src))
NAMES = {"spot_bunny_phong_1080p": "3 spot + bunny PHONG 1920×1080", "spot_x16_texture_2048": "4 spot ×16 TEXTURE 2048²",
         "spot_x8_overdraw_4096": "5 overdraw ×8 NORMAL/PHONG 4096²", "readme_spot_crate_1024": "README scene: spot + Crate1.obj 1024²",
         "spot_texture_1024_3lights": "2 with 3 lights", "spot_texture_1024_p32": "2 with p = 32",
         "spot_texture_1024_p7.5": "2 with p = 7.5 (`pow_fast` builds)", "spot_bump_1024": "2 with the BUMP shader"}


def row(name, e, top=False):
    r, v = e["roofline"], e.get("valu") or {}
    fps = e["value"] if top else e["frames_per_sec"]
    toa, vf, vp, o = r.get("traffic_over_algorithmic"), v.get("valu_frac"), v.get("valu_pipe_frac_est"), r["one_stream"]
    return (f"| {name} | {e['config']['frames_per_step'] if top else e['frames_per_step']}, {r.get('lanes', 1)} | {fps:,.0f} | {e['ms_per_step']:.3f} | "
            f"{e['mfragments_per_sec'] / 1e3:.1f} | **{r['frac']:.3f}** / {r['frac'] * 8000 / 6290:.3f} | {'' if vf is None else f'{vf:.2f} / {vp:.2f}'} | {'' if toa is None else f'{toa:.2f}'} | "
            f"{o['k_setup_bin_ms'] * 1e3:.0f} / {o['k_raster_ms'] * 1e3:.0f} / {o['k_shade_ms'] * 1e3:.0f} |")


def driver_table():
    recs = sorted(glob.glob(os.path.join(REPO, "BENCH_r??.json")))
    if not recs:
        return "(no driver record BENCH_rNN.json at the repo root yet)\n"
    rec = json.load(open(recs[-1]))
    p, nn = rec.get("parsed"), 'r' + str(int(os.path.basename(recs[-1])[7:9]))
    if not p:
        return f"(the driver's record {os.path.basename(recs[-1])} holds no parsed line)\n"
    r = p["roofline"]
    cells = [("2 (`value`)", p["value"], p["ms_per_step"], r["frac"])] + [(k[1:], r.get("fps_" + k), None, r.get("frac_" + k)) for k in ("c3", "c4", "c5")]
    t = [f"**driver** — `{os.path.basename(recs[-1])}` (`{rec.get('cmd', 'python bench.py --steps 20 --warmup 5')}`, run by the driver on its own box at the end of round {nn[1:]}, "
         f"for the code of round {nn[1:]}; `driver_run_s` {rec.get('driver_run_s')}; kernel hash `{r.get('kernel_source_hash')}`):", "",
         "| config | frames/s | ms/step | frac of 8.0 TB/s | frac of 6.29 TB/s measured |", "|---|---|---|---|---|"]
    for name, fps, ms, fr in cells:
        if fr is not None:
            t.append(f"| {name} | {fps:,.0f} | {'' if ms is None else f'{ms:.3f}'} | **{fr:.3f}** | {fr * 8000 / 6290:.3f} |")
    cb = p.get("cpu_baseline") or {}
    t.append("")
    t.append(f"(unprimed region: frac {r.get('frac_unprimed')}; `cpu_baseline` {cb.get('value', 0):,.0f} {cb.get('unit', '')} on {cb.get('cores')} threads, kind `{cb.get('kind')}`)")
    return "\n".join(t) + "\n"


out = [driver_table(), f"**builder** — `profiles/{os.path.basename(src)}` (the same command on a box of the pool, this round's code):", "",
       "| workload | frames/step, lanes | frames/s | ms/step | Gfragments/s | frac of 8.0 TB/s / of 6.29 TB/s measured | `valu_frac` / `valu_pipe_frac_est` | traffic ÷ algorithmic | one stream: setup+bin / raster / shade µs |",
       "|---|---|---|---|---|---|---|---|---|", row("2 spot TEXTURE 1024² (`value`)", d, True)]
loop = None
for e in d["configs"]:
    if e.get("scope") == "readme_loop":
        loop = e
    elif "error" not in e:
        nm = NAMES.get(e["workload"]) or ("2, scope `draw` (vertex stage timed)" if e["scope"] == "draw" else "2, one frameset on one stream")
        if e.get("approx_shade"):
            nm = (NAMES.get(e["workload"]) or "2 spot TEXTURE 1024²").split(" (")[0] + " — tolerance mode"
        out.append(row(nm, e))
txt = "\n".join(out)
txt += (f"\n\n(`python bench.py --steps {d['steps']} --warmup {d['warmup']}`, `profiles/{os.path.basename(src)}`; `value` is the second timed region; the first one, "
        f"straight after the {d['warmup']} warm-up steps from idle: {d['value_unprimed']:,.0f} frames/s / {d['roofline']['algorithmic_bytes_per_launch'] / (d['ms_per_step_unprimed'] * 1e-3) / 8e12:.3f}.  "
        "Driver records of earlier rounds: round 5 278 830 frames/s / 0.620, configs 3 / 4 / 5 0.598 / 0.398 / 0.414; round 4 (builder's run) 287 000 / 0.641, 0.593 / 0.406 / 0.415; round 3 247 200 / 0.549, 0.484 / 0.338 / 0.340.)  "
        "The reference's own protocol through the C++ API "
        f"(`readme_loop`): `draw()` until the device has finished **{loop['draw_complete_ms']['median']:.3f} ms** (p10 {loop['draw_complete_ms']['p10']:.3f} / "
        f"p90 {loop['draw_complete_ms']['p90']:.3f}; submit {loop['draw_submit_ms']['median']:.3f}), `display()` incl. the 8-bit resolve and the 3 MB read-back "
        f"{loop['display_ms']['median']:.3f} ms; the reference publishes 17.06 ms for `draw()` on an i7-12800HX.  `cpu_baseline`: "
        f"{d['cpu_baseline']['value']:.0f} frames/s on {d['cpu_baseline']['cores']} threads.")
p = os.path.join(REPO, "DESIGN.md")
s = open(p).read()
s = re.sub(r"<!-- numbers:begin -->.*?<!-- numbers:end -->", "<!-- numbers:begin -->\n" + txt + "\n<!-- numbers:end -->", s, flags=re.S)
open(p, "w").write(s)
print(txt)

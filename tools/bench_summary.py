"""dev tool: one line per case of a bench.py JSON line (stdin or file)"""
import json, sys
txt = open(sys.argv[1]).read() if len(sys.argv) > 1 else sys.stdin.read()
d = json.loads(txt.strip().splitlines()[-1])
r = d["roofline"]
print(f"{d['config']['workload']:26s} {d['config']['scope']:6s} {d['value']:9.0f} f/s {d['ms_per_step']:.4f} ms frac {r['frac']:.4f}  one-stream {r['one_stream']['ms_per_step']:.4f}  p10/50/90 {[round(x,3) for x in d['ms_per_step_p10_median_p90']]}")
for e in d.get("configs", []):
    if "error" in e:
        print("   ", e); continue
    if e.get("scope") == "readme_loop":
        print(f"{e['workload']:26s} readme_loop  draw_complete p10/50/90 {e['draw_complete_ms']['p10']:.4f} / {e['draw_complete_ms']['median']:.4f} / {e['draw_complete_ms']['p90']:.4f} ms"
              f"  submit {e['draw_submit_ms']['median']:.4f}  display {e['display_ms']['median']:.4f}"); continue
    r = e["roofline"]
    print(f"{e['workload']:26s} {e['scope']:6s} lanes {e.get('lanes', 1)} {e['frames_per_sec']:9.0f} f/s {e['ms_per_step']:.4f} ms frac {r['frac']:.4f}  one-stream {r['one_stream']['ms_per_step']:.4f}")

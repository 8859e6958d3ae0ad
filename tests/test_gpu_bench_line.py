"""-m gpu: the default bench.py line (N = 1) as the driver runs it — one JSON line whose `roofline.per_config` carries every
BASELINE GPU config, the README scene, the scope `draw`, the non-integer-exponent variant and the reference's own published
protocol (`readme_loop`), whose `cpu_baseline` is measured in the same run, and whose PMC constants are either those of THIS
tree's kernel sources or flagged stale."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_default_line_carries_every_config_in_the_part_the_driver_keeps():
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "SRZ_BENCH_FORCE_LAUNCHER")}
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--steps", "3", "--warmup", "1", "--cpu-budget-s", "2"],
                       capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith('{"metric"')]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["metric"] == "frames_per_sec" and d["unit"] == "frames/s" and d["n_gpus"] == 1 and d["steps"] == 3 and d["warmup"] == 1
    assert d["higher_is_better"] is True and d["scaling"] == "weak" and d["dtype"] == "f32" and d["vs_baseline"] is None
    assert d["config"]["workload"] == "spot_texture_1024" and d["config"]["frames_per_step"] == 256 and d["priming_steps"] == 19
    roof = d["roofline"]
    assert roof["bound"] == "hbm" and roof["peak"] == 8000.0 and 0.2 < roof["frac"] < 1.0 and abs(roof["frac"] - roof["achieved"] / roof["peak"]) < 1e-9
    assert (roof["traffic"] is None) == bool(roof.get("traffic_stale"))          # fresh counters, or flagged — never silently old ones
    pc = roof["per_config"]
    for w in ("spot_bunny_phong_1080p", "spot_x16_texture_2048", "spot_x8_overdraw_4096", "readme_spot_crate_1024",
              "spot_texture_1024_p7.5", "spot_texture_1024:draw"):
        assert "error" not in pc[w], pc[w]
        assert pc[w]["frames_per_sec"] > 0 and 0.05 < pc[w]["frac"] < 1.0 and pc[w]["one_stream_us"]["shade"] > 0
    loop = pc["readme_spot_crate_1024:readme_loop"]
    assert "error" not in loop, loop
    p10, med, p90 = loop["draw_complete_ms_p10_median_p90"]
    assert 0 < p10 <= med <= p90 < 5.0 and loop["frames"] == 1000 and loop["reference_published_draw_ms_median"] == 17.06
    cb = d["cpu_baseline"]
    assert cb["kind"] == "port" and cb["value"] > 0 and cb["cores"] >= 1 and cb["unit"] == "frames/s" and "sample" in cb
    assert any(e.get("scope") == "readme_loop" for e in d["configs"]) and len(d["configs"]) >= 11

"""bench.py's own launcher (no GPU needed): `python bench.py --gpus N` without torchrun around it must start N fresh ranks
BEFORE touching torch / HIP, relay their failure and never hang; and the PMC constants of the bench line are tied to the kernel
sources they were collected from."""
import json
import os
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_bench(args, env_extra=None, timeout=240):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "HIP_VISIBLE_DEVICES")}
    env["HIP_VISIBLE_DEVICES"] = ""  # a box with a GPU behaves like one without: the ranks must refuse, not compute
    env["CUDA_VISIBLE_DEVICES"] = ""
    env.update(env_extra or {})
    return subprocess.run([sys.executable, os.path.join(REPO, "bench.py")] + args, capture_output=True, text=True, timeout=timeout, env=env)


def test_gpus_2_spawns_two_ranks_and_fails_loudly_without_a_gpu():
    r = run_bench(["--gpus", "2", "--steps", "1", "--warmup", "0"])
    assert r.returncode != 0
    # both ranks were started (torchrun's report names them) and each said why it cannot run
    assert r.stderr.count("needs an MI355X") >= 1, r.stderr[-2000:]
    assert "local_rank: 1" in r.stderr or "rank      : 1" in r.stderr, r.stderr[-2000:]
    assert '{"metric"' not in r.stdout  # no line without a measurement


def test_forced_launcher_at_one_rank_takes_the_same_path():
    r = run_bench(["--gpus", "1", "--steps", "1", "--warmup", "0"], {"SRZ_BENCH_FORCE_LAUNCHER": "1"})
    assert r.returncode != 0 and "needs an MI355X" in r.stderr


def test_launcher_runs_before_torch_is_imported():
    """the parent must not import torch (let alone initialise HIP) before it starts the ranks"""
    src = open(os.path.join(REPO, "bench.py")).read()
    main = src[src.index("def main():"):]
    assert main.index("launch_ranks(") < main.index("import torch")
    head = src[:src.index("def main():")]
    assert "\nimport torch" not in head and "\nfrom torch" not in head


def test_pmc_counters_are_tied_to_the_library_build(tmp_path, monkeypatch):
    sys.path.insert(0, REPO)
    import bench
    h = bench.kernel_source_hash()
    assert h.startswith("fb-") and len(h) == 19 and h == bench.kernel_source_hash()  # (the library's .hip_fatbin section)
    monkeypatch.setattr(bench, "_PMC", {"_kernel_source_hash": h, "w": {"frames_per_step": 4, "hbm_bytes_per_step": 1.0}})
    assert not bench.pmc_stale() and bench.pmc_counters("w", 4, "raster")["hbm_bytes_per_step"] == 1.0
    monkeypatch.setattr(bench, "_PMC", {"_kernel_source_hash": "fb-" + "0" * 16, "w": {"frames_per_step": 4, "hbm_bytes_per_step": 1.0}})
    assert bench.pmc_stale() and bench.pmc_counters("w", 4, "raster") is None
    # the committed file: either collected from this tree's sources, or reported as stale — never silently another version's
    monkeypatch.setattr(bench, "_PMC", None)
    committed = json.load(open(os.path.join(REPO, "profiles", "pmc_counters.json")))
    assert bench.pmc_stale() == (committed.get("_kernel_source_hash") != h)


def test_multi_gpu_budget_of_the_default_eight_rank_run_fits_hbm():
    sys.path.insert(0, REPO)
    sys.path.insert(0, os.path.join(REPO, "software-rasterizer_amd"))
    import argparse
    import bench
    b = bench.multi_gpu_budget(argparse.Namespace(frames=256), 8, "spot_texture_1024")
    assert b["frames_per_step"] == 2048
    assert b["gathered_buffers_bytes_planes"] == 2 * 8 * 2048 * 16 * 128 * 1024     # 2 x world x shard
    assert b["shard_bytes_bgr8"] * 16 == b["shard_bytes_planes"] * 3
    assert b["peak_bytes_estimate"] < 0.85 * b["hbm_bytes"]
    p = bench.predicted_exchange(b["shard_bytes_bgr8"], b["frames_per_step"])
    assert 5.0 < p["exchange_ms_at_xgmi_peak"] < 5.6 and 3.6e5 < p["frames_per_sec_if_exchange_bound"] < 4.1e5


def test_emit_prints_one_short_last_line_and_keeps_the_full_record_beside_it(tmp_path, monkeypatch, capsys):
    """the compact line stays below LINE_LIMIT even when fed round 4's full 21 KB record (the one the driver could not parse), is
    the last thing on stdout, carries configs 3 / 4 / 5 as flat scalars of `roofline`, and the full record lands in the details file"""
    sys.path.insert(0, REPO)
    import bench
    full = json.load(open(os.path.join(REPO, "profiles", "r04_bench_driver_args.json")))
    extras = full.pop("configs")
    full["value_unprimed"], full["ms_per_step_unprimed"] = 270000.0, 0.948
    n8 = {"shard_render_ms": [0.123456] * 8, "max_over_mean": 1.03, "predicted_step_ms": {"planes": 3.5, "bgr8": 0.66},
          "predicted_fps": {"planes": 1.0e4, "bgr8": 2.0e4}}
    full["roofline"]["multi_gpu_emulated"] = {"spot_texture_1024": {"frames_per_gpu": 32, "N2": n8, "N4": n8, "N8": n8},
                                              "spot_x16_texture_2048": {"frames_per_gpu": 8, "N8": n8},
                                              "spot_x8_overdraw_4096": {"frames_per_gpu": 4, "N8": n8}}
    for k in ("spot_texture_1024:approx", "spot_bunny_phong_1080p:approx", "spot_x8_overdraw_4096:approx"):
        full["roofline"]["per_config"][k] = dict(full["roofline"]["per_config"]["spot_bunny_phong_1080p"])
    monkeypatch.setattr(bench, "DETAILS", str(tmp_path / "bench_details.json"))
    bench.emit(full, extras)
    out = capsys.readouterr().out
    lines = [ln for ln in out.splitlines() if ln.strip()]
    assert len(lines) == 1 and len(lines[0]) < bench.LINE_LIMIT <= 6000, len(lines[0])
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
              "data", "config", "roofline", "cpu_baseline", "value_unprimed", "ms_per_step_unprimed"):
        assert k in d, k
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["peak"] == 8000.0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3 and "traffic" in r
    for c in ("c3", "c4", "c5"):
        assert r["frac_" + c] == r["per_config"][c]["frac"] and r["fps_" + c] > 0
    assert len(r["multi_gpu_emulated"]["c2"]["N8"]["ms"]) == 8 and len(d["cpu_baseline"]["sample"]) <= 200
    kept = json.load(open(tmp_path / "bench_details.json"))
    assert len(kept["configs"]) == len(extras) >= 11 and kept["value"] == full["value"]
    # a record that is too long still yields a parseable line: the nested tables go first, the flat scalars stay
    full["roofline"]["per_config"] = {f"w{i}": dict(full["roofline"]["per_config"]["spot_bunny_phong_1080p"]) for i in range(80)}
    bench.emit(full, extras)
    line = capsys.readouterr().out.strip().splitlines()[-1]
    assert len(line) < bench.LINE_LIMIT and "frac" in json.loads(line)["roofline"]

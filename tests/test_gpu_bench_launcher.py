"""-m gpu: bench.py's launcher path end to end at ONE rank — `SRZ_BENCH_FORCE_LAUNCHER=1 python bench.py --gpus 1` starts one
child through torch.distributed.run, which runs exactly the code an N > 1 rank runs (init_process_group("nccl"), the collective
communicator set-up, time_multi_gpu with both exchanges) at world 1 — so that the 8-GPU scaling run is not that code's first
execution — and its JSON line has the schema of the direct N = 1 line."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ARGS = ["--gpus", "1", "--steps", "3", "--warmup", "1", "--frames", "16", "--no-cpu-baseline", "--no-extras"]


def run(env_extra):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env.update(env_extra)
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py")] + ARGS, capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, (r.returncode, r.stderr[-3000:])
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith('{"metric"')]
    assert len(lines) == 1, r.stdout[-2000:]
    return json.loads(lines[0])


def test_launcher_path_at_one_rank_runs_the_multi_gpu_branch():
    direct = run({})
    forced = run({"SRZ_BENCH_FORCE_LAUNCHER": "1"})
    assert "multi_gpu" not in direct and direct["n_gpus"] == 1
    m = forced["multi_gpu"]
    assert forced["n_gpus"] == 1 and forced["value"] > 0 and forced["steps"] == 3
    assert m["headline_exchange"] == "planes" and m["behind_c_abi"] and m["overlapped"] and not m["second_pass"]
    assert m["bgr8"]["frames_per_sec"] > 0 and m["bgr8"]["bytes_sent_per_rank_per_step"] == 16 * 3 * 1024 * 1024
    assert m["bytes_sent_per_rank_per_step"] == 16 * 16 * 1024 * 1024
    assert m["predicted"]["bgr8"]["exchange_ms_at_xgmi_peak"] > 0
    assert "fallback" not in m, m.get("fallback")           # the RCCL communicator behind the C ABI was really built
    # same schema as the direct line (+ multi_gpu)
    assert set(direct) <= set(forced) | {"cpu_baseline"}
    for k in ("metric", "unit", "higher_is_better", "scaling", "dtype", "data", "vs_baseline"):
        assert direct[k] == forced[k]
    assert set(direct["config"]) == set(forced["config"]) and set(direct["roofline"]) <= set(forced["roofline"])
    assert forced["config"]["sharding"].startswith("32-row bands round-robin over 1 GPUs + RCCL all-gather of planes")

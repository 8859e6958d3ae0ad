"""-m gpu: the C++ host surface end to end on the device — the reference README's usage (tests/cpp/api_demo.cpp) compiled
against our headers must run to completion: device vertex stage == host vertex stage bit for bit, device 8-bit resolve ==
host rounding of the float planes, stats consistent with the z-buffer (the checks are inside the program); the planes of
its last frame are compared with the oracle here."""
import os
import subprocess

import pytest

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_readme_program_runs_on_the_gpu(tmp_path, orc):
    exe = tmp_path / "api_demo"
    cmd = ["g++", "-std=c++17", "-O1", os.path.join(REPO, "tests", "cpp", "api_demo.cpp"), "-I",
           os.path.join(REPO, "software-rasterizer_amd", "host", "include"), "-L", os.path.join(REPO, "software-rasterizer_amd"),
           "-lsrz_host", "-lsrz", f"-Wl,-rpath,{os.path.join(REPO, 'software-rasterizer_amd')}", "-o", str(exe)]
    subprocess.check_call(cmd)
    dump = tmp_path / "planes.f32"
    r = subprocess.run([str(exe), REPO, str(dump)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, (r.returncode, r.stdout, r.stderr)
    assert "covered=" in r.stdout
    # its last frame (spot rotated by 130 degrees = frame 13 of the 10-degree sequence) against the oracle, bit for bit
    import numpy as np
    import scenes
    got = np.fromfile(dump, np.float32).reshape(4, 256, 256)
    rc, ref, _ = orc.draw(scenes.config2(13, size=256))
    assert rc == 0
    for p in range(4):
        assert np.array_equal(got[p].view(np.uint32), np.ascontiguousarray(ref[p]).view(np.uint32)), p

"""CPU only: the host layer's file parsers and the oracle under AddressSanitizer + UndefinedBehaviourSanitizer.

The PNG / PPM / JPEG decoders and the OBJ reader of software-rasterizer_amd/host stand in for cv::imread and tinyobjloader
(/root/reference src/TextureLoader.cpp:3-12, src/ObjLoader.cpp:197-233) and parse files a user hands them.  `make -C
software-rasterizer_amd asan` builds them with -fsanitize=address,undefined around tests/cpp/fuzz_host.cpp, which feeds seeded
byte- and structure-level mutations of fixture files through the loaders' entry points: every file must load or raise the
documented std::runtime_error — any sanitizer report, other exception or crash fails the run.  `make -C oracle asan` does the same
for the checker itself (oracle/srz_oracle.c) on seeded adversarial frames.

SRZ_FUZZ_FILES = mutations per seed file (default 400; with ~5 seeds per format that is ~2 000 files per format).  One-off runs of
this harness with SRZ_FUZZ_FILES=20000 are recorded in NOTEBOOK.md (round 6)."""
import os
import subprocess

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(REPO, "software-rasterizer_amd")
N = int(os.environ.get("SRZ_FUZZ_FILES", "400"))
ENV = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0:allocator_may_return_null=1", UBSAN_OPTIONS="print_stacktrace=1")


@pytest.fixture(scope="module")
def fuzz_host():
    subprocess.check_call(["make", "-s", "-C", PKG, "asan"])
    return os.path.join(PKG, "build", "fuzz_host_asan")


def run(exe, kind, seed_file, n, seed, tmp):
    r = subprocess.run([exe, kind, seed_file, str(n), str(seed), str(tmp)], capture_output=True, text=True, timeout=1200, env=ENV)
    assert r.returncode == 0, f"{kind} {os.path.basename(seed_file)} seed {seed}: rc {r.returncode}\n{r.stdout[-500:]}\n{r.stderr[-4000:]}"
    ok, rej = (int(x.split("=")[1]) for x in r.stdout.split())
    assert ok + rej == n and ok >= 1
    return ok, rej


def test_jpeg_decoder_survives_mutated_files(fuzz_host, tmp_path):
    d = os.path.join(REPO, "tests", "golden", "jpeg")
    seeds = ["base_420_odd.jpg", "base_420_restart.jpg", "base_444_64.jpg", "prog_420_odd.jpg", "prog_444_noise.jpg", "prog_grey.jpg", "tiny_3x2_prog.jpg"]
    tot = [0, 0]
    for i, s in enumerate(seeds):
        ok, rej = run(fuzz_host, "jpg", os.path.join(d, s), N, 100 + i, tmp_path)
        tot[0] += ok; tot[1] += rej
    assert tot[0] > N and tot[1] > N // 4, tot      # both outcomes are exercised: files that still decode, files that are refused


def test_png_and_ppm_decoders_survive_mutated_files(fuzz_host, tmp_path):
    from PIL import Image
    rng = np.random.default_rng(5)
    a = rng.integers(0, 256, (13, 17, 4), dtype=np.uint8)
    files = {"rgb.png": Image.fromarray(a[:, :, :3]), "rgba.png": Image.fromarray(a), "l.png": Image.fromarray(a[:, :, 0]),
             "p.png": Image.fromarray(a[:, :, 0]).convert("P"), "la.png": Image.fromarray(np.ascontiguousarray(a[:, :, :2])),
             "l16.png": Image.fromarray(a[:, :, 0].astype(np.uint16) * 257)}
    tot = [0, 0]
    for i, (name, im) in enumerate(files.items()):
        p = str(tmp_path / name)
        im.save(p)
        ok, rej = run(fuzz_host, "png", p, N, 200 + i, tmp_path)
        tot[0] += ok; tot[1] += rej
    ok, rej = run(fuzz_host, "png", os.path.join(REPO, "assets", "models", "Crate", "Crate1.png"), max(20, N // 20), 299, tmp_path)
    assert tot[0] > N // 2 and tot[1] > N, tot
    for i, (name, im) in enumerate((("rgb.bmp", Image.fromarray(a[:, :, :3])), ("rgba.bmp", Image.fromarray(a)),
                                    ("pal.bmp", Image.fromarray(a[:, :, :3]).quantize(19)))):
        q = str(tmp_path / name)
        im.save(q)
        run(fuzz_host, "bmp", q, N, 250 + i, tmp_path)
    p = tmp_path / "s.ppm"
    p.write_bytes(b"P6\n# c\n17 13\n255\n" + a[:, :, :3].tobytes())
    run(fuzz_host, "ppm", str(p), N, 300, tmp_path)


def test_obj_reader_survives_mutated_files(fuzz_host, tmp_path):
    crate = os.path.join(REPO, "assets", "models", "Crate", "Crate1.obj")
    tot = [0, 0]
    ok, rej = run(fuzz_host, "obj", crate, 3 * N, 400, tmp_path)
    tot[0] += ok; tot[1] += rej
    # a short file with every index form, negative indices, a quad and a polygon, no normals
    small = tmp_path / "small.obj"
    small.write_text("o m\nv 0 0 0\nv 1 0 0 1 0 0\nv 1 1 0\nv 0 1 0\nv 0.5 0.5 1\nvt 0 0\nvt 1 0\nvt 1 1\nvn 0 0 1\n"
                     "f 1/1/1 2/2/1 3/3/1\nf 1//1 3//1 4//1\nf 1/1 2/2 5/3\nf -1 -2 -3\nf 1 2 3 4\nf 1 2 3 4 5\ng tail\n")
    ok, rej = run(fuzz_host, "obj", str(small), 2 * N, 401, tmp_path)
    tot[0] += ok; tot[1] += rej
    assert tot[0] > N and tot[1] >= 1, tot


def test_oracle_reads_and_writes_inside_its_buffers(tmp_path):
    subprocess.check_call(["make", "-s", "-C", os.path.join(REPO, "oracle"), "asan"])
    exe = os.path.join(REPO, "oracle", "build", "oracle_asan")
    for seed in (1, 2, 3):
        r = subprocess.run([exe, str(max(20, N // 10)), str(seed)], capture_output=True, text=True, timeout=1200, env=dict(ENV, OMP_NUM_THREADS="2"))
        assert r.returncode == 0 and r.stdout.startswith("frames="), r.stdout[-300:] + r.stderr[-4000:]

"""Generates tests/golden/*.json|npz from the CPU oracle (python tests/golden/make_golden.py).

The reference holds no golden vectors and cannot be built here (parity unpinned), so these fixtures pin the ORACLE
against regressions and give the GPU tests fixed answers: per-plane bit checksums, the resolved 8-bit image hash,
counters, and ~1k sampled pixels (x, y, z, c0, c1, c2) per case."""
import hashlib
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import conftest  # noqa: F401,E402
import scenes  # noqa: E402
from oracle import oracle  # noqa: E402

CASES = {
    "config1_256": lambda: scenes.config1(),
    "config2_1024_f0": lambda: scenes.config2(0),
    "config2_1024_f7": lambda: scenes.config2(7),
    "config2_1024_f0_phong": lambda: scenes.config2(0, shader=2),
    "config2_1024_f0_normal": lambda: scenes.config2(0, shader=0),
    "config3_1080p_f3": lambda: scenes.config3(3),
    "config4_2048_f2": lambda: scenes.config4(2),
    "config5_1024_f1": lambda: scenes.config5(1, size=1024),
}


def digest(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def summarize(planes, stats):
    z, c0, c1, c2 = planes
    cov = np.isfinite(z)
    ys, xs = np.nonzero(cov)
    rng = np.random.default_rng(12345)
    sel = rng.choice(len(xs), size=min(1000, len(xs)), replace=False) if len(xs) else np.zeros(0, int)
    samples = np.stack([xs[sel], ys[sel], z[ys[sel], xs[sel]].view(np.uint32), c0[ys[sel], xs[sel]].view(np.uint32),
                        c1[ys[sel], xs[sel]].view(np.uint32), c2[ys[sel], xs[sel]].view(np.uint32)], 1).astype(np.uint32)
    meta = {"stats": stats, "sha256": {"z": digest(z), "c0": digest(c0), "c1": digest(c1), "c2": digest(c2),
                                       "bgr8": digest(oracle.resolve8(planes))},
            "covered": int(cov.sum()), "sum_z": float(z[cov].astype(np.float64).sum()),
            "sum_c": [float(p.astype(np.float64).sum()) for p in (c0, c1, c2)]}
    return meta, samples


def main():
    oracle.texture_set(scenes.TEX_SPOT, scenes.spot_texture())
    out, samples = {}, {}
    for name, make in CASES.items():
        f = make()
        rc, planes, st = oracle.draw(f)
        assert rc == 0
        out[name], samples[name] = summarize(planes, st)
        print(name, st)
    json.dump(out, open(os.path.join(HERE, "golden.json"), "w"), indent=1, sort_keys=True)
    np.savez_compressed(os.path.join(HERE, "golden_samples.npz"), **samples)


if __name__ == "__main__":
    main()

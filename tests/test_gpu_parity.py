"""-m gpu: the HIP path (through the C ABI) against the CPU oracle on the same inputs.

Tolerances (stated here as required):
  z          bit-exact expected (same IEEE ops in the same order); asserted: max |Δz| <= 1e-5*|z| and the count of
             non-identical z values is reported and bounded (<= 1e-4 of pixels)
  coverage   identical owner set expected; pixels where only one side is covered must be <= 1e-4 of pixels
  colour     |Δc| <= 0.5 on the 0..255 scale everywhere; scalar-tail pixels are integers on both sides; the number of
             pixels with any colour difference > 1e-3 must be <= 1e-3 of covered pixels
  8-bit      resolved image >= 99.9 % identical
"""
import numpy as np
import pytest

import scenes
from srz import abi

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    import srz
    c = srz.Context(0)
    c.texture_upload(scenes.TEX_SPOT, scenes.spot_texture())
    yield c
    c.close()


def compare(gpu, ref, name):
    gz, g0, g1, g2 = gpu
    rz, r0, r1, r2 = ref
    n = gz.size
    gcov, rcov = np.isfinite(gz), np.isfinite(rz)
    cov_diff = int((gcov != rcov).sum())
    both = gcov & rcov
    z_ne = int((gz[both].view(np.uint32) != rz[both].view(np.uint32)).sum())
    zrel = float(np.max(np.abs(gz[both] - rz[both]) / np.abs(rz[both]))) if both.any() else 0.0
    dc = np.maximum.reduce([np.abs(g0 - r0), np.abs(g1 - r1), np.abs(g2 - r2)])
    dc_max = float(dc[both].max()) if both.any() else 0.0
    n_c = int((dc[both] > 1e-3).sum())
    print(f"[{name}] pixels={n} covered={int(rcov.sum())} coverage_diff={cov_diff} z_not_bit_identical={z_ne} "
          f"max_rel_dz={zrel:.3g} max_dcolour={dc_max:.4g} colour_diff_gt_1e-3={n_c}")
    assert cov_diff <= 1e-4 * n
    assert z_ne <= 1e-4 * n and zrel <= 1e-5
    assert dc_max <= 0.5
    assert n_c <= 1e-3 * max(1, int(rcov.sum()))
    return dict(cov_diff=cov_diff, z_ne=z_ne, dc_max=dc_max, n_c=n_c)


def run_both(ctx, orc, frame, want_stats=True):
    rc, ref, rst = orc.draw(frame)
    assert rc == 0
    gpu, gst = ctx.draw(frame, want_stats=want_stats)
    return gpu, ref, gst, rst


def test_config1_flat_triangles(ctx, orc):
    f = scenes.config1()
    gpu, ref, gst, rst = run_both(ctx, orc, f)
    r = compare(gpu, ref, "config1")
    assert r["cov_diff"] == 0 and r["z_ne"] == 0
    assert gst == rst
    assert gpu[1][100, 128] == 127.5 and gpu[2][100, 128] == 127.5 and gpu[3][100, 128] == 0.0


@pytest.mark.parametrize("shader", [abi.SHADER_NORMAL, abi.SHADER_TEXTURE, abi.SHADER_PHONG])
@pytest.mark.parametrize("frame_idx", [0, 7])
def test_config2_spot_1024(ctx, orc, shader, frame_idx):
    f = scenes.config2(frame_idx, shader=shader)
    gpu, ref, gst, rst = run_both(ctx, orc, f)
    compare(gpu, ref, f"config2 shader={shader} frame={frame_idx}")
    assert gst == rst
    g8, r8 = orc.resolve8(gpu), orc.resolve8(ref)
    assert (g8 == r8).mean() >= 0.999

"""-m gpu: the HIP path (through the C ABI) against the CPU oracle on the same inputs.

Bar (north_star: "within a stated per-channel float tolerance, z-buffer bit-exact where integer"):
the kernels perform the oracle's IEEE binary32 operations in the oracle's order (no contraction, correctly rounded
div/sqrt, pow evaluated in binary64), so the tests assert BIT-IDENTICAL z, coverage, counters and — for every shader —
colour planes.  The only place the two sides may legitimately differ is pow(): both evaluate it in binary64 and round
once, through different binary64 algorithms (glibc pow vs square-and-multiply / ocml pow); a last-place difference of
the binary64 value flips the binary32 rounding with probability ~1e-7 per evaluation.  The stated tolerance for that case:
    |Δcolour| <= 1e-3 on the 0..255 scale on at most 1e-5 of the covered pixels; everything else bit-identical.
"""
import numpy as np
import pytest

import scenes
from srz import abi
from test_golden import check_against_golden
from test_oracle_kat import ccw, frame, tri

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    import srz
    c = srz.Context(0)
    c.texture_upload(scenes.TEX_SPOT, scenes.spot_texture())
    yield c
    c.close()


def bits(a):
    return np.ascontiguousarray(a).view(np.uint32)


def compare(gpu, ref, name):
    gz, rz = gpu[0], ref[0]
    n_cov = max(1, int(np.isfinite(rz).sum()))
    z_same = np.array_equal(bits(gz), bits(rz))
    dc = np.maximum.reduce([np.abs(g.astype(np.float64) - r.astype(np.float64)) for g, r in zip(gpu[1:], ref[1:])])
    dc = np.nan_to_num(dc, nan=0.0) + np.where(np.isnan(gpu[1]) != np.isnan(ref[1]), 1e9, 0.0)
    n_diff = int(sum((bits(g) != bits(r)).sum() for g, r in zip(gpu[1:], ref[1:])))
    print(f"[{name}] covered={n_cov} z_bit_identical={z_same} colour_values_not_bit_identical={n_diff} max_dcolour={dc.max():.3g}")
    assert z_same, f"{name}: z-buffer is not bit-identical"
    assert dc.max() <= 1e-3 and (dc > 0).sum() <= max(1, int(1e-5 * n_cov)), f"{name}: colour outside the stated tolerance"
    return n_diff


def run_both(ctx, orc, f, planes_init=None, want_stats=True):
    def clone():
        return None if planes_init is None else tuple(p.copy() for p in planes_init)
    rc, ref, rst = orc.draw(f, clone())
    assert rc == 0
    gpu, gst = ctx.draw(f, clone(), want_stats=want_stats)
    if want_stats:
        assert gst == rst, (gst, rst)
    return gpu, ref


# ------------------------------------------------------------------------------------------------ BASELINE configs
def test_config1_flat_triangles(ctx, orc):
    gpu, ref = run_both(ctx, orc, scenes.config1())
    assert compare(gpu, ref, "config1") == 0
    assert (gpu[1][100, 128], gpu[2][100, 128], gpu[3][100, 128]) == (127.5, 127.5, 0.0)
    check_against_golden("config1_256", gpu)


@pytest.mark.parametrize("shader", [abi.SHADER_NORMAL, abi.SHADER_TEXTURE, abi.SHADER_PHONG])
@pytest.mark.parametrize("frame_idx", [0, 7, 20])
def test_config2_spot_1024(ctx, orc, shader, frame_idx):
    gpu, ref = run_both(ctx, orc, scenes.config2(frame_idx, shader=shader))
    compare(gpu, ref, f"config2 shader={shader} frame={frame_idx}")
    assert (orc.resolve8(gpu) == orc.resolve8(ref)).all()


def test_config2_against_committed_golden(ctx):
    gpu, st = ctx.draw(scenes.config2(0), want_stats=True)
    check_against_golden("config2_1024_f0", gpu, st, exact=False)
    gpu, st = ctx.draw(scenes.config2(7), want_stats=True)
    check_against_golden("config2_1024_f7", gpu, st, exact=False)


def test_config3_spot_bunny_1080p(ctx, orc):
    gpu, ref = run_both(ctx, orc, scenes.config3(3))
    compare(gpu, ref, "config3")
    check_against_golden("config3_1080p_f3", gpu, exact=False)


def test_config4_spot_x16_2048(ctx, orc):
    gpu, ref = run_both(ctx, orc, scenes.config4(2))
    compare(gpu, ref, "config4")


def test_config5_overdraw_4096(ctx, orc):
    gpu, ref = run_both(ctx, orc, scenes.config5(1))
    compare(gpu, ref, "config5 4096")


# ------------------------------------------------------------------------------------------------ flags / accumulation
@pytest.mark.parametrize("frame_idx", [0, 5, 21])
def test_readme_scene_spot_and_crate(ctx, orc, frame_idx):
    """the scene of the reference's published raster timing (README.md:619-642): spot + Crate1.obj, two textures"""
    orc.texture_set(scenes.TEX_CRATE, scenes.crate_texture())
    ctx.texture_upload(scenes.TEX_CRATE, scenes.crate_texture())
    gpu, ref = run_both(ctx, orc, scenes.readme_scene(frame_idx))
    compare(gpu, ref, f"readme scene frame={frame_idx}")
    assert (orc.resolve8(gpu) == orc.resolve8(ref)).all()
    f = scenes.readme_scene(frame_idx)
    rc, spot_only, _ = orc.draw(abi.Frame(f.width, f.height, scenes.README_EYE, scenes.LIGHTS, [(abi.SHADER_TEXTURE, scenes.TEX_SPOT, f.tris[0])],
                                          abi.FUSED_CLEAR))
    crate_px = int(np.isfinite(ref[0]).sum() - np.isfinite(spot_only[0]).sum())
    assert crate_px > 2000, crate_px                        # the 12 crate triangles do cover their share of the frame


def test_unified_flag(ctx, orc):
    gpu, ref = run_both(ctx, orc, scenes.config2(5, size=512, flags=abi.FUSED_CLEAR | abi.UNIFIED))
    compare(gpu, ref, "unified")


def test_draw_accumulates_without_clear(ctx, orc):
    """draw() never clears: two scenes drawn one after the other == both batches in one frame (composition property)."""
    a, b = scenes.config2(2, size=512, flags=0), scenes.config2(11, size=512, flags=0, shader=abi.SHADER_PHONG)
    init = orc.new_planes(512, 512)
    init[0][100:300, 200:260] = 85.0     # a pre-existing occluder in the z-buffer, with its own colour
    init[2][100:300, 200:260] = 33.0
    rc, ref, _ = orc.draw(a, tuple(p.copy() for p in init))
    rc, ref, _ = orc.draw(b, ref)
    gpu, _ = ctx.draw(a, tuple(p.copy() for p in init))
    gpu, _ = ctx.draw(b, gpu)
    compare(gpu, ref, "accumulate a then b")
    both = abi.Frame(512, 512, scenes.EYE, scenes.LIGHTS, [(abi.SHADER_TEXTURE, 0, a.tris[0]), (abi.SHADER_PHONG, -1, b.tris[0])], 0)
    gpu2, _ = ctx.draw(both, tuple(p.copy() for p in init))
    for x, y in zip(gpu, gpu2):
        assert np.array_equal(bits(x), bits(y))
    assert (gpu[2][110:120, 210:220] == 33.0).any()      # occluder colour survives where it is nearer


def test_idempotent_redraw(ctx, orc):
    """Drawing the same scene again over its own output changes nothing (V: z<z fails; S: <= passes, same values)."""
    f = scenes.config2(9, size=512, flags=0)
    gpu, _ = ctx.draw(f)
    again, _ = ctx.draw(f, tuple(p.copy() for p in gpu))
    for x, y in zip(gpu, again):
        assert np.array_equal(bits(x), bits(y))


# ------------------------------------------------------------------------------------------------ edge cases
@pytest.mark.parametrize("w,h", [(101, 67), (250, 130), (33, 31), (1, 1), (4, 40)])
def test_odd_sizes(ctx, orc, w, h):
    t = np.concatenate([ccw((w * 0.1, h * 0.1), (w * 0.95, h * 0.2), (w * 0.3, h * 0.9)),
                        ccw((-5, -5), (w + 9.5, 3), (2, h + 7.25), z=60.0)])
    for flags in (abi.FUSED_CLEAR, 0):
        gpu, ref = run_both(ctx, orc, frame(t, w=w, h=h, flags=flags))
        compare(gpu, ref, f"odd {w}x{h} flags={flags}")


def test_empty_frames(ctx, orc):
    f = abi.Frame(64, 64, (0, 0, 1), np.zeros((0, 2, 3), np.float32), [], abi.FUSED_CLEAR)
    gpu, st = ctx.draw(f, want_stats=True)
    assert not np.isfinite(gpu[0]).any() and (gpu[1] == 0).all() and st["n_tris"] == 0
    f = frame(np.zeros(0, abi.TRI_DTYPE))
    gpu, st = ctx.draw(f, want_stats=True)
    assert not np.isfinite(gpu[0]).any()
    z = np.full((64, 64), 3.0, np.float32)
    c = np.full((64, 64), 9.0, np.float32)
    gpu, _ = ctx.draw(frame(np.zeros(0, abi.TRI_DTYPE), flags=0), (z, c.copy(), c.copy(), c.copy()))
    assert (gpu[0] == 3.0).all() and (gpu[1] == 9.0).all()      # nothing drawn, nothing cleared


def test_degenerate_offscreen_and_nonfinite_triangles(ctx, orc):
    t = np.concatenate([ccw((10, 10), (10, 10), (10, 10)),                 # zero area
                        ccw((5, 5), (40, 40), (22.5, 22.5)),               # collinear
                        ccw((-50, 10), (-10, 10), (-30, 40)),              # off-screen (collapses to column 0)
                        ccw((1e30, 0), (0, 1e30), (-1e30, -1e30)),         # huge finite
                        ccw((8, 8), (28.5, 8), (8, 28.5)),                 # a real one
                        ccw((-1000, -1000), (3000, -1000), (-1000, 3000), z=70.0)])  # covers the whole screen
    t2 = t.copy()
    t2["pos"][4][1][2] = np.inf                                            # non-finite → dropped
    for tt in (t, t2):
        gpu, ref = run_both(ctx, orc, frame(tt, eye=(0, 0, 1)))
        compare(gpu, ref, "degenerate")


@pytest.mark.parametrize("shader", [abi.SHADER_PHONG, abi.SHADER_TEXTURE, abi.SHADER_NORMAL])
def test_operands_outside_the_fast_math_range(ctx, orc, shader):
    """k_shade's optimistic short rcp/sqrt sequences must hand over to the IEEE expansions: zero / huge / tiny normals
    (sqrt of 0, of +inf, of a denormal), a light straight above a pixel (1/sqrt(0)) and a light at the eye."""
    tris = np.concatenate([
        ccw((4, 4), (30.5, 4), (4, 30.5), nrm=(0, 0, 0)),                                   # |n| = 0
        ccw((34, 4), (60.5, 4), (34, 30.5), nrm=(1e30, -1e30, 1e30)),                       # |n|^2 overflows
        ccw((4, 34), (30.5, 34), (4, 60.5), nrm=(1e-30, 1e-30, -1e-30)),                    # |n|^2 underflows
        ccw((34, 34), (47.5, 34), (34, 47.5), nrm=(0.3, -0.2, -1)),                         # ordinary, V + S columns
        ccw((50, 50), (55, 50), (50, 55), nrm=(0, 0, -1), uv=((0.1, 0.1), (0.9, 0.2), (0.4, 0.8)))])  # S columns only
    lights = [[(40.0, 40.0, 60.0), (500, 500, 500)],   # x, y = a pixel corner of the fourth triangle: lx = ly = 0
              [(0.0, 0.0, 1.0), (300, 200, 100)]]      # at the eye
    f = frame(tris, shader=shader, lights=lights, tex=scenes.TEX_SPOT if shader == abi.SHADER_TEXTURE else -1)
    gpu, ref = run_both(ctx, orc, f)
    compare(gpu, ref, f"fast-math fallback, shader {shader}")
    assert ctx.debug_counters()[11] >= 1, "no tile took the IEEE pass: the test does not exercise the fallback"
    f = frame(tris, shader=shader, lights=lights, tex=scenes.TEX_SPOT if shader == abi.SHADER_TEXTURE else -1, flags=abi.FUSED_CLEAR | abi.UNIFIED)
    gpu, ref = run_both(ctx, orc, f)
    compare(gpu, ref, f"fast-math fallback (unified), shader {shader}")


def test_z_ties_v_first_wins_s_last_wins(ctx, orc):
    a = ccw((8, 8), (28.5, 8), (8, 28.5))
    b = a.copy()
    b["nrm"][0] = [[0, 0, 1]] * 3
    gpu, ref = run_both(ctx, orc, frame(np.concatenate([a, b, a, b])))
    compare(gpu, ref, "z ties")
    assert gpu[3][9, 12] == 0.0 and gpu[3][9, 25] == 255.0


@pytest.mark.parametrize("p", [150.0, 0.0, 1.0, 7.5, 64.0, 3000.0])
def test_specular_exponent_variants(ctx, orc, p):
    f = scenes.config2(4, size=256, shader=abi.SHADER_PHONG)
    f.c.p = p
    gpu, ref = run_both(ctx, orc, f)
    compare(gpu, ref, f"p={p}")


@pytest.mark.parametrize("n_lights", [0, 1, 5])
def test_light_counts(ctx, orc, n_lights):
    rng = np.random.default_rng(n_lights)
    L = np.concatenate([rng.uniform(-1, 1, (n_lights, 1, 3)), rng.uniform(10, 120, (n_lights, 1, 3))], 1).astype(np.float32)
    base = scenes.config2(6, size=256)
    f = abi.Frame(256, 256, scenes.EYE, L, [(abi.SHADER_TEXTURE, 0, base.tris[0])], abi.FUSED_CLEAR)
    gpu, ref = run_both(ctx, orc, f)
    compare(gpu, ref, f"{n_lights} lights")


@pytest.mark.parametrize("shader", [abi.SHADER_BUMP, abi.SHADER_DISPLACEMENT])
def test_bump_and_displacement(ctx, orc, shader):
    gpu, ref = run_both(ctx, orc, scenes.config2(3, size=512, shader=shader))
    compare(gpu, ref, f"shader {shader}")
    assert (gpu[1] == 255.0).any()       # the 8-wide columns are the reference's empty SIMD stubs: white


@pytest.mark.parametrize("shader", [abi.SHADER_BUMP, abi.SHADER_DISPLACEMENT])
def test_bump_with_the_height_map_the_reference_ships(ctx, orc, shader):
    """examples/models/spot/hmap.jpg (a progressive JPEG, copied to assets/) is what the reference's bump / displacement shaders
    are meant to sample: decoded by the C++ host layer's own JPEG decoder (libjpeg-turbo's bytes: tests/test_host_layer.py),
    uploaded as texture 5, the same bytes given to the oracle"""
    import os
    from srz import host
    hm = host.load_image_bgr(os.path.join(os.path.dirname(scenes.SPOT_OBJ), "hmap.jpg"))
    assert hm.shape == (800, 800, 3)
    orc.texture_set(5, hm)
    ctx.texture_upload(5, hm)
    f0 = scenes.config2(6, size=512)
    f = abi.Frame(512, 512, scenes.EYE, scenes.LIGHTS, [(shader, 5, f0.tris[0])], abi.FUSED_CLEAR)
    gpu, ref = run_both(ctx, orc, f)
    compare(gpu, ref, f"hmap.jpg shader {shader}")


def test_small_texture_and_uv_clamps(ctx, orc):
    rng = np.random.default_rng(3)
    tex = rng.integers(0, 256, (5, 7, 3), dtype=np.uint8)
    orc.texture_set(9, tex)
    ctx.texture_upload(9, tex)
    t = ccw((2, 2), (60.5, 3), (4, 61.5))
    t["uv"][0] = [[-0.3, 1.0], [1.0, 0.2], [1.4, -0.1]]
    f = frame(t, shader=abi.SHADER_TEXTURE, tex=9, lights=scenes.LIGHTS)
    gpu, ref = run_both(ctx, orc, f)
    compare(gpu, ref, "tiny texture")


def test_many_batches_and_shader_mix(ctx, orc):
    base = scenes.config2(8, size=512).tris[0]
    parts = np.array_split(base, 70)      # more batches than the LDS descriptor cache ever held
    shaders = [abi.SHADER_NORMAL, abi.SHADER_TEXTURE, abi.SHADER_PHONG]
    f = abi.Frame(512, 512, scenes.EYE, scenes.LIGHTS, [(shaders[i % 3], 0, p) for i, p in enumerate(parts)], abi.FUSED_CLEAR)
    gpu, ref = run_both(ctx, orc, f)
    compare(gpu, ref, "70 batches")


def test_primitive_and_texture_errors(ctx):
    import srz
    t = ccw((8, 8), (28.5, 8), (8, 28.5))
    gpu, _ = ctx.draw(frame(t), primitive=abi.PRIMITIVE_LINES)      # LINES is accepted and drawn filled
    assert np.isfinite(gpu[0]).any()
    with pytest.raises(srz.SrzError) as e:
        ctx.draw(frame(t), primitive=7)
    assert e.value.code == abi.SRZ_E_PRIMITIVE and "Primitive Type is not supported!" in str(e.value)
    with pytest.raises(srz.SrzError) as e:
        ctx.draw(frame(t, shader=abi.SHADER_TEXTURE, tex=55))
    assert e.value.code == abi.SRZ_E_TEXTURE


def test_deterministic(ctx):
    f = scenes.config2(13)
    a, _ = ctx.draw(f)
    b, _ = ctx.draw(f)
    for x, y in zip(a, b):
        assert np.array_equal(bits(x), bits(y))

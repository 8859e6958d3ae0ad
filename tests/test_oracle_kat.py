"""not-gpu: hand-derived known-answer tests that pin the CPU oracle to the reference's semantics
(SURVEY.md §8a "two semantics" table and Appendix B).  The reference holds no fixtures of its own (parity unpinned)."""
import numpy as np
import pytest

from srz import abi

F32 = np.float32


def tri(a, b, c, z=50.0, nrm=(0, 0, -1), uv=((0, 0), (0, 0), (0, 0))):
    t = np.zeros(1, abi.TRI_DTYPE)
    za, zb, zc = (z, z, z) if np.isscalar(z) else z
    t["pos"][0] = [[a[0], a[1], za], [b[0], b[1], zb], [c[0], c[1], zc]]
    t["nrm"][0] = [nrm] * 3 if np.ndim(nrm) == 1 else nrm
    t["uv"][0] = uv
    return t


def frame(tris, w=64, h=64, shader=abi.SHADER_NORMAL, eye=(0, 0, 1), lights=(), flags=abi.FUSED_CLEAR, tex=-1, **kw):
    batches = tris if isinstance(tris, list) else [(shader, tex, tris)]
    return abi.Frame(w, h, eye, np.asarray(lights, np.float32).reshape(-1, 2, 3), batches, flags, **kw)


# ---------------------------------------------------------------------------------------------- matrices (glm)
def test_perspective_lh_no_with_raw_45(orc):
    p = orc.perspective_lh_no(45.0, 1.0, 0.1, 100.0).reshape(4, 4)  # [col][row]
    th = np.tan(F32(45.0) / F32(2.0), dtype=F32)
    assert abs(float(th) - 0.5578) < 2e-3  # tan(22.5 rad): the fovy-in-degrees quirk (src/Scene.cpp:293)
    assert p[0, 0] == F32(1.0) / (F32(1.0) * th) and p[1, 1] == F32(1.0) / th
    assert p[2, 2] == (F32(100.0) + F32(0.1)) / (F32(100.0) - F32(0.1))
    assert p[2, 3] == 1.0 and p[3, 2] == -(F32(2.0) * F32(100.0) * F32(0.1)) / (F32(100.0) - F32(0.1))
    assert p[3, 3] == 0.0 and np.count_nonzero(p) == 5


def test_look_at_lh(orc):
    v = orc.look_at_lh((0, 0, 0.9), (0, 0, 0), (0, 1, 0)).reshape(4, 4)
    # f=(0,0,-1), s=cross(up,f)=(-1,0,0), u=cross(f,s)=(0,1,0); translation = (-s.e, -u.e, -f.e) = (0,0,0.9)
    expect = np.array([[-1, 0, 0, 0], [0, 1, 0, 0], [0, 0, -1, 0], [0, 0, F32(0.9), 1]], F32)
    assert np.array_equal(v, expect)  # -0.0 == 0.0


def test_ndc_matrix_stretches_x_by_aspect(orc):
    n = orc.ndc_matrix(1920, 1080).reshape(4, 4)
    aspect = F32(1920) / F32(1080)
    assert n[0, 0] == F32(960.0) * aspect and n[1, 1] == 540.0 and n[3, 0] == 960.0 and n[3, 1] == 540.0
    assert n[2, 2] == 1.0 and n[3, 3] == 1.0


def test_model_matrix_is_T_R_S(orc):
    m = orc.model_matrix((0, 1, 0), 90.0, (1, 2, 3), (2, 2, 2)).reshape(4, 4).astype(np.float64)
    # rotate 90 deg about +y (glm::rotate): x-axis -> -z, z-axis -> +x ; scale 2 ; translate
    expect = np.array([[0, 0, -2, 0], [0, 2, 0, 0], [2, 0, 0, 0], [1, 2, 3, 1]], np.float64)
    assert np.allclose(m, expect, atol=2e-7)


def test_inverse_and_mul(orc):
    m = orc.model_matrix((1, 2, 3), 37.0, (0.3, -0.2, 0.5), (0.3, 0.4, 0.5))
    prod = orc.m4_mul(m, orc.m4_inverse(m)).reshape(4, 4)
    assert np.allclose(prod, np.eye(4), atol=1e-5)
    t = orc.m4_transpose(m).reshape(4, 4)
    assert np.array_equal(t, m.reshape(4, 4).T)
    v = orc.m4_mulv(m, (1, 2, 3, 1))
    ref = m.reshape(4, 4).T.astype(np.float64) @ np.array([1, 2, 3, 1.0])
    assert np.allclose(v, ref, rtol=1e-6)


def test_vertex_stage_depth_remap_and_normal_w(orc):
    verts = np.array([[0, 0, 0, 0, 0, 1, 0.25, 0.75], [1, 0, 0, 0, 0, 1, 0, 0], [0, 1, 0, 0, 0, 1, 0, 0]], F32)
    faces = np.array([[0, 1, 2]], np.uint32)
    W = H = 256
    view = orc.look_at_lh((0, 0, 0.9), (0, 0, 0), (0, 1, 0))
    proj = orc.perspective_lh_no(45.0, 1.0, 0.1, 100.0)
    model = orc.model_matrix((0, 1, 0), 0.0, (0, 0, 0), (1, 1, 1))
    t = orc.vertex_stage(verts, faces, model, view, proj, orc.ndc_matrix(W, H), 0.1, 100.0)[0]
    # origin maps to the screen centre; z_view = 0.9 → NDC z = (A*0.9+B)/0.9, remapped to [near,far]
    assert t["pos"][0][0] == 128.0 and t["pos"][0][1] == 128.0
    A, B = (100.1 / 99.9), -(2 * 100 * 0.1) / 99.9
    zn = (A * 0.9 + B) / 0.9
    assert abs(t["pos"][0][2] - (zn * 49.95 + 50.05)) < 1e-3
    assert 0.1 <= t["pos"][0][2] <= 100.0
    assert tuple(t["uv"][0]) == (0.25, 0.75)
    assert np.allclose(t["nrm"][0], (0, 0, 1))  # identity model: (M^-1)^T n with w=1, divided by w=1


# ---------------------------------------------------------------------------------------------- bbox / cull
def test_bbox_truncates_toward_zero_then_clamps(orc):
    # x in [-3.7, 5.9] → trunc → [-3, 5] → clamp [0,5]; rows [2.2, 9.9] → [2, 9]; all columns < 8 wide-chunk → S class
    t = tri((-3.7, 2.2), (5.9, 2.2), (1.0, 9.9))
    rc, _, st = orc.draw(frame(t, eye=(0, 0, -1)))
    assert rc == 0 and st["pixel_tests"] == 6 * 8 and st["n_culled"] == 0


def test_offscreen_triangle_collapses_to_an_edge_column(orc):
    t = tri((-50, 10), (-10, 10), (-30, 40))  # entirely left of the screen: bbox clamps to column 0
    rc, pl, st = orc.draw(frame(t, eye=(0, 0, -1)))
    assert st["pixel_tests"] == 1 * 31 and st["fragments"] == 0 and not np.isfinite(pl[0]).any()


def test_backface_uses_eye_position(orc):
    t = tri((10, 10), (50, 10), (30, 50))  # cross((40,0,0),(20,40,0)) = +z
    assert orc.draw(frame(t, eye=(0, 0, 1)))[2]["n_culled"] == 1   # dot(n, eye) = +1 > 0 → culled
    assert orc.draw(frame(t, eye=(0, 0, -1)))[2]["n_culled"] == 0
    assert orc.draw(frame(t, eye=(0, 0, 0)))[2]["n_culled"] == 0   # dot == 0 is NOT > 0


def test_nonfinite_vertex_is_dropped(orc):
    t = tri((10, 10), (50, 10), (30, 50))
    t["pos"][0][1][0] = np.nan
    rc, pl, st = orc.draw(frame(t, eye=(0, 0, -1)))
    assert rc == 0 and st["n_culled"] == 1 and not np.isfinite(pl[0]).any()


# ---------------------------------------------------------------------------------------------- coverage
def ccw(a, b, c, **kw):  # helper: a winding that survives the cull for eye=(0,0,1)
    return tri(a, c, b, **kw)


def test_pixels_on_an_edge_are_not_covered_in_either_class(orc):
    # V class: legs of 32 → the area term is -1024, its reciprocal and every barycentric are exact in binary32, so the
    # strict compares (0 < a,b,c < 1) exclude exactly the on-edge pixels.  bbox x 10..42 = 33 columns → V for x < 42.
    t = ccw((10, 10), (42, 10), (10, 42))
    rc, pl, st = orc.draw(frame(t))
    cov = np.isfinite(pl[0])
    assert not cov[:, 10].any() and not cov[10, :].any()
    assert cov[11, 11] and cov[11, 40] and not cov[11, 41]     # hypotenuse x + y = 52
    assert cov[30, 21] and not cov[30, 22]
    assert int(cov.sum()) == sum(max(0, 52 - y - 11) for y in range(11, 42))
    # S class: bbox 6 columns wide → every column is scalar-tail; edge functions on integers are exact
    t = ccw((10, 10), (15, 10), (10, 15))
    rc, pl, st = orc.draw(frame(t))
    ys, xs = np.nonzero(np.isfinite(pl[0]))
    assert sorted(zip(xs.tolist(), ys.tolist())) == [(11, 11), (11, 12), (11, 13), (12, 11), (12, 12), (13, 11)]


def test_sample_point_is_the_pixel_corner_not_the_centre(orc):
    # hypotenuse x+y = 24.5: integer corners with x,y >= 11 and x+y <= 24 are inside; the pixel (10,10), whose CENTRE
    # (10.5,10.5) is inside, is not: the +0.5 of src/Rasterizer.cpp:465 is lost in the size_t parameter
    t = ccw((10.25, 10.25), (14.25, 10.25), (10.25, 14.25))
    rc, pl, st = orc.draw(frame(t))
    ys, xs = np.nonzero(np.isfinite(pl[0]))
    assert sorted(zip(xs.tolist(), ys.tolist())) == [(11, 11), (11, 12), (11, 13), (12, 11), (12, 12), (13, 11)]


def test_v_columns_vs_scalar_tail_split(orc):
    # bbox x 8..28 = 21 columns → V for x in [8,24), S for x in [24,28]
    t = ccw((8, 8), (28.5, 8), (8, 28.5))
    rc, pl, _ = orc.draw(frame(t))
    c0 = pl[1]
    assert c0[9, 12] == 127.5            # V class: float colour clamp01((0+1)*0.5)*255
    assert c0[9, 25] == 127.0            # S class: normalizedToRGB truncates to an unsigned integer
    assert pl[3][9, 12] == 0.0 and pl[3][9, 25] == 0.0


def test_unified_flag_makes_every_column_v(orc):
    t = ccw((8, 8), (28.5, 8), (8, 28.5))
    rc, pl, _ = orc.draw(frame(t, flags=abi.FUSED_CLEAR | abi.UNIFIED))
    assert pl[1][9, 25] == 127.5


# ---------------------------------------------------------------------------------------------- z-test
def test_z_tie_first_wins_in_v_columns_last_wins_in_scalar_tail(orc):
    a = ccw((8, 8), (28.5, 8), (8, 28.5))
    b = a.copy()
    a["nrm"][0] = [[0, 0, -1]] * 3   # colour (127.5,127.5,0)
    b["nrm"][0] = [[0, 0, 1]] * 3    # colour (127.5,127.5,255)
    rc, pl, st = orc.draw(frame(np.concatenate([a, b])))
    assert pl[3][9, 12] == 0.0       # V: z < zbuf is strict → the FIRST triangle keeps the pixel
    assert pl[3][9, 25] == 255.0     # S: !(z > zbuf) → the LAST triangle overwrites
    assert st["shaded"] > st["visible"]


def test_nearer_triangle_wins_regardless_of_order(orc):
    near = ccw((8, 8), (28.5, 8), (8, 28.5))
    far = near.copy()
    near["pos"][0][:, 2] = 40.0
    far["pos"][0][:, 2] = 60.0
    near["nrm"][0] = [[0, 0, 1]] * 3
    for order in ([near, far], [far, near]):
        rc, pl, _ = orc.draw(frame(np.concatenate(order)))
        assert pl[0][9, 12] == 40.0 and pl[0][9, 25] == 40.0 and pl[3][9, 12] == 255.0


def test_draw_never_clears_without_the_flag(orc):
    t = ccw((8, 8), (28.5, 8), (8, 28.5))
    z, c0, c1, c2 = orc.new_planes(64, 64)
    z[:] = 45.0
    c2[:] = 7.0
    rc, pl, _ = orc.draw(frame(t, flags=0), (z, c0, c1, c2))   # z=50 is behind 45 → nothing drawn
    assert (pl[0] == 45.0).all() and (pl[3] == 7.0).all()
    z[:] = 55.0
    rc, pl, _ = orc.draw(frame(t, flags=0), (z, c0, c1, c2))
    assert pl[0][9, 12] == 50.0 and pl[0][40, 40] == 55.0 and pl[3][40, 40] == 7.0


# ---------------------------------------------------------------------------------------------- texture / shading
def test_texture_fetch_semantics(orc):
    tex = np.zeros((4, 4, 3), np.uint8)
    tex[:, :, 0] = np.arange(16).reshape(4, 4) * 10        # "blue" channel carries the texel id
    tex[:, :, 1] = 255
    orc.texture_set(5, tex)
    lights = [((20, 20, 0), (0, 0, 0))]                     # zero intensity → colour = 0, only exercise the fetch path
    t = ccw((8, 8), (28.5, 8), (8, 28.5), )
    t["uv"][0] = [[1.0, 1.0]] * 3
    rc, pl, _ = orc.draw(frame(t, shader=abi.SHADER_TEXTURE, tex=5, lights=lights))
    assert rc == 0
    # with I=0 everything is 0; use ka*I path: give intensity via a second run
    lights = [((20, 20, 0), (1e4, 1e4, 1e4))]
    rc, pl, _ = orc.draw(frame(t, shader=abi.SHADER_TEXTURE, tex=5, lights=lights))
    # V class: u*texW clamped to texW-1 = 3 → texel (3,3) = id 15 → blue = 150/255 → saturates to 255 under I=1e4
    assert pl[1][9, 12] == 255.0 and pl[2][9, 12] == 255.0 and pl[3][9, 12] == 0.0   # red channel of tex is 0
    # S class: u == 1.0 → index == size → BLACK (src/TextureLoader.cpp:25-27)
    assert pl[1][9, 25] == 0.0 and pl[2][9, 25] == 0.0 and pl[3][9, 25] == 0.0


def test_missing_texture_is_an_error(orc):
    t = ccw((8, 8), (28.5, 8), (8, 28.5))
    rc, _, _ = orc.draw(frame(t, shader=abi.SHADER_TEXTURE, tex=63))
    assert rc == abi.SRZ_E_TEXTURE


def test_phong_known_value_v_class(orc):
    """One light straight 'above' the pixel in screen space: closed-form Blinn-Phong of the reference's mixed-space model."""
    t = ccw((8, 8), (28.5, 8), (8, 28.5), z=50.0)
    t["nrm"][0] = [[0, 0, 1]] * 3
    L = ((12.0, 9.0, 60.0), (10.0, 20.0, 30.0))
    rc, pl, _ = orc.draw(frame(t, shader=abi.SHADER_PHONG, lights=[L], eye=(0, 0, 1)))
    y, z = 9.0, 50.0
    # pixel (12,9): light_dir = (0,0,10) → att = 1/sqrt(0) = inf; the specular term is inf*0 = NaN and
    # _mm256_max_ps(NaN, 0) = 0 (second operand) → the reference stores 0 there
    assert pl[1][9, 12] == 0.0
    x = 14.0   # pixel (14,9): light_dir = (-2,0,10), att = 1/2
    l = np.array([12.0 - x, 0.0, 10.0])
    att = 1.0 / np.hypot(l[0], l[1])
    n = np.array([0, 0, 1.0])
    cosA = max(0.0, (l / np.linalg.norm(l)) @ n)
    h = l + (np.array([0, 0, 1.0]) - np.array([x, y, z]))
    cosT = max(0.0, (h / np.linalg.norm(h)) @ n) ** 150
    for ch, I in enumerate((10.0, 20.0, 30.0)):
        c = 1.0 * (0.005 * I + (I * att * 1.0) * cosA + (I * att * 0.7937) * cosT)
        expect = min(max(c, 0.0), 1.0) * 255.0
        assert abs(pl[1 + ch][9, 14] - expect) < 1e-2


def test_bump_and_displacement_are_white_in_v_columns(orc):
    tex = np.full((4, 4, 3), 128, np.uint8)
    orc.texture_set(6, tex)
    t = ccw((8, 8), (28.5, 8), (8, 28.5))
    for sh in (abi.SHADER_BUMP, abi.SHADER_DISPLACEMENT):
        rc, pl, _ = orc.draw(frame(t, shader=sh, tex=6, lights=[((0, 0, 0), (1, 1, 1))]))
        assert rc == 0 and pl[1][9, 12] == 255.0 and pl[2][9, 12] == 255.0 and pl[3][9, 12] == 255.0   # SIMD stubs


# ---------------------------------------------------------------------------------------------- draw() contract
def test_primitive_validation(orc):
    t = ccw((8, 8), (28.5, 8), (8, 28.5))
    assert orc.draw(frame(t), primitive=abi.PRIMITIVE_LINES)[0] == 0          # LINES draws filled triangles
    assert orc.draw(frame(t), primitive=7)[0] == abi.SRZ_E_PRIMITIVE


def test_resolve8_rounds_half_to_even_and_saturates(orc):
    c = np.array([[0.5, 1.5, 2.5, 254.5, 255.5, 300.0, -3.0, np.nan]], F32)
    out = orc.resolve8((None, c, c, c))
    assert out[0, :, 0].tolist() == [0, 2, 2, 254, 255, 255, 0, 0]


def test_config1_plumbing(orc):
    import scenes
    rc, pl, st = orc.draw(scenes.config1())
    assert rc == 0 and st["n_tris"] == 3 and st["n_culled"] == 0
    assert (pl[1][100, 128], pl[2][100, 128], pl[3][100, 128]) == (127.5, 127.5, 0.0)
    assert pl[0][100, 128] == 50.0 and pl[0][150, 150] == 40.0   # the z=40 triangle wins where it overlaps
    assert st["fragments"] > st["shaded"] >= st["visible"] > 0


def test_omp_and_row_band_draws_equal_the_serial_draw(orc):
    import scenes
    f = scenes.config2(3, size=256)
    rc, ref, _ = orc.draw(f)
    pl = orc.new_planes(256, 256)
    rc, n = orc.draw_omp(f, pl, band=8)
    assert rc == 0 and n >= 1
    for a, b in zip(pl, ref):
        assert np.array_equal(a.view(np.uint32), b.view(np.uint32))


# ---------------------------------------------------------------------------------------------- closed forms from the source, in numpy binary64
def _bary(P, A, B, C):
    """alpha, beta, gamma of the pixel corner P as src/Rasterizer.cpp:53-70 / :89-127 define them: areas of (P,B,C) and (P,C,A) over (A,B,C)"""
    cr = lambda u, v: u[0] * v[1] - u[1] * v[0]  # noqa: E731
    area = cr(B - A, C - A)
    al, be = cr(B - P, C - P) / area, cr(C - P, A - P) / area
    return al, be, 1.0 - al - be


def _blinn_phong64(P, n, kd, eye, lights, ka=0.005, ks=0.7937, p=150.0):
    """Shader::BlinnPhong (src/Shader.cpp:510-543) summed over the lights, per channel, before the clamp"""
    n = n / np.linalg.norm(n)
    out = np.zeros(3)
    for Lp, I in lights:
        Lp, I = np.asarray(Lp, float), np.asarray(I, float)
        l = Lp - P
        d = I / np.sqrt((Lp[0] - P[0]) ** 2 + (Lp[1] - P[1]) ** 2)          # "distanceSquared" is a 2-D distance (quirk 7)
        cos_t = max(0.0, n @ (l / np.linalg.norm(l)))
        h = l + (np.asarray(eye, float) - P)
        cos_a = max(0.0, n @ (h / np.linalg.norm(h)))
        out += (ka * I + cos_t * kd * d + cos_a ** p * ks * d) * kd
    return out


def test_phong_known_value_scalar_tail_class(orc):
    """a pixel of the scalar-tail columns, two coloured lights, a slanted normal: Shader::BlinnPhong in closed form, clamped, times 255,
    truncated (Tools::normalizedToRGB, src/Tools.cpp:94-104)"""
    A, B, C = np.array([8.0, 8.0]), np.array([8.0, 28.5]), np.array([28.5, 8.0])   # the record's vertex order (tri = ccw with b, c swapped)
    t = tri(tuple(A), tuple(B), tuple(C), z=(40.0, 55.0, 70.0))
    t["nrm"][0] = [[0.2, -0.3, 0.9]] * 3
    lights = [((40.0, 3.0, 90.0), (6.0, 9.0, 12.0)), ((5.0, 30.0, 20.0), (3.0, 2.0, 1.0))]
    eye = (0.0, 0.0, 1.0)
    rc, pl, _ = orc.draw(frame(t, shader=abi.SHADER_PHONG, lights=lights, eye=eye))
    assert rc == 0
    for (x, y) in ((25, 9), (26, 9), (24, 11)):                       # bbox x 8..28: columns 24..28 are the scalar tail
        P2 = np.array([float(x), float(y)])
        al, be, ga = _bary(P2, A, B, C)
        assert min(al, be, ga) > 0
        z = al * 40.0 + be * 55.0 + ga * 70.0
        assert abs(pl[0][y, x] - z) < 1e-4
        c = _blinn_phong64(np.array([x, y, z], float), np.array([0.2, -0.3, 0.9]), np.ones(3), eye, lights)
        want = np.floor(np.clip(c, 0.0, 1.0) * 255.0)
        got = np.array([pl[1][y, x], pl[2][y, x], pl[3][y, x]])
        assert (got == np.floor(got)).all() and np.abs(got - want).max() <= 1.0, (x, y, got, want, np.clip(c, 0, 1) * 255)
        assert 5.0 < got.min() and got.max() < 250.0 and len(set(got.tolist())) == 3   # a meaningful, unsaturated, coloured value


def test_interpolated_normals_and_texture_coordinates_in_both_classes(orc):
    """per-vertex normals and uv: NORMAL shader = the interpolated, normalised normal as a colour (V: float, S: truncated);
    TEXTURE shader: the texel the interpolated uv selects (V: round half even of u*w clamped to w-1, include/loader/TextureLoader.hpp:26-74;
    S: truncation of clamp(u)*w, src/TextureLoader.cpp:14-31) — the texel's id is read back through an ambient-only light"""
    A, B, C = np.array([8.0, 8.0]), np.array([8.0, 28.5]), np.array([28.5, 8.0])   # the record's vertex order
    nrm = np.array([[0.0, 0.0, -1.0], [0.6, 0.0, -0.8], [0.0, 0.8, -0.6]])
    uv = np.array([[0.05, 0.10], [0.95, 0.15], [0.10, 0.90]])
    t = tri(tuple(A), tuple(B), tuple(C))
    t["nrm"][0], t["uv"][0] = nrm, uv
    rc, pl, _ = orc.draw(frame(t))
    assert rc == 0
    for (x, y, s_class) in ((12, 9, False), (20, 12, False), (25, 9, True), (24, 11, True)):
        al, be, ga = _bary(np.array([float(x), float(y)]), A, B, C)
        n = al * nrm[0] + be * nrm[1] + ga * nrm[2]
        n = n / np.linalg.norm(n)
        want = (n + 1.0) / 2.0 * 255.0
        got = np.array([pl[1][y, x], pl[2][y, x], pl[3][y, x]])
        if s_class:
            assert (got == np.floor(got)).all() and np.abs(got - np.floor(want)).max() <= 1.0, (x, y, got, want)
        else:
            assert np.abs(got - want).max() < 2e-3, (x, y, got, want)
    # texel ids: an 8 x 8 texture whose blue channel is 4 * (row * 8 + col); with intensity I and ka the colour is kd * ka * I * 255
    tex = np.zeros((8, 8, 3), np.uint8)
    tex[:, :, 0] = (np.arange(64).reshape(8, 8) * 4).astype(np.uint8)
    orc.texture_set(7, tex)
    far = [((1e6, 1e6, 1e6), (40.0, 40.0, 40.0))]      # a light so far away that only the ambient term ka * I is left (d ~ 1e-5)
    rc, pl, _ = orc.draw(frame(t, shader=abi.SHADER_TEXTURE, tex=7, lights=far))
    assert rc == 0
    for (x, y, s_class) in ((12, 9, False), (20, 12, False), (25, 9, True), (24, 11, True)):
        al, be, ga = _bary(np.array([float(x), float(y)]), A, B, C)
        u, v = al * uv[0] + be * uv[1] + ga * uv[2]
        if s_class:
            col, row = int(min(max(u, 0.0), 1.0) * 8), int(min(max(v, 0.0), 1.0) * 8)
        else:
            col, row = int(np.rint(min(max(u * 8, 0.0), 7.0))), int(np.rint(min(max(v * 8, 0.0), 7.0)))
        kd = tex[row, col, 0] / 255.0
        want = (0.005 * 40.0) * kd * 255.0            # Shader::BlinnPhong returns (La + Ld + Ls) * colour with La = ka * I (no kd inside)
        got = pl[1][y, x]
        assert abs(got - (np.floor(want) if s_class else want)) <= (1.0 if s_class else 0.05), (x, y, s_class, row, col, got, want)


def test_bump_scalar_tail_known_value(orc):
    """calcBumpMapping + BlinnPhong in closed form for a scalar-tail pixel (src/Shader.cpp:477-507, 624-640): the perturbed normal
    TBN * (-dU, -dV, 1) from the lengths of three texel colours, then the Blinn-Phong sum with kd = the texel"""
    A, B, C = np.array([8.0, 8.0]), np.array([8.0, 28.5]), np.array([28.5, 8.0])
    rng = np.random.default_rng(11)
    tex = rng.integers(40, 216, (8, 8, 3)).astype(np.uint8)
    orc.texture_set(8, tex)
    nvec = np.array([0.3, 0.5, -0.81])
    t = tri(tuple(A), tuple(B), tuple(C), z=30.0)
    t["nrm"][0] = [nvec] * 3
    t["uv"][0] = [[0.30, 0.40]] * 3
    lights = [((60.0, 40.0, -30.0), (9.0, 9.0, 9.0))]
    eye = (0.0, 0.0, 1.0)
    rc, pl, _ = orc.draw(frame(t, shader=abi.SHADER_BUMP, tex=8, lights=lights, eye=eye))
    assert rc == 0
    x, y = 25, 9

    def texel(u, v):                                   # TextureLoader::getTextureColor(vec2): clamp, truncate, black at the far edge
        cu, cv = min(max(u, 0.0), 1.0), min(max(v, 0.0), 1.0)
        cx, cy = int(cu * 8), int(cv * 8)
        return np.zeros(3) if (cx >= 8 or cy >= 8) else tex[cy, cx].astype(float) / 255.0
    n = nvec / np.linalg.norm(nvec)                    # the rasteriser hands the shader glm::normalize(interpolated normal)
    s = np.sqrt(n[0] ** 2 + n[2] ** 2)
    tv = np.array([n[0] * n[1] / s, s, n[2] * n[1] / s])
    bv = np.cross(n, tv)
    kh, kn = 0.2, 0.1
    on = np.linalg.norm(texel(0.30, 0.40))
    dU = kh * kn * (np.linalg.norm(texel((0.30 + 1) / 8, 0.40)) - on)
    dV = kh * kn * (np.linalg.norm(texel(0.30, (0.40 + 1) / 8)) - on)
    ln = np.array([-dU, -dV, 1.0])
    nb = np.array([tv @ ln, bv @ ln, n @ ln])
    nb = nb / np.linalg.norm(nb)
    c = _blinn_phong64(np.array([x, y, 30.0], float), nb, texel(0.30, 0.40), eye, lights)
    want = np.floor(np.clip(c, 0.0, 1.0) * 255.0)
    got = np.array([pl[1][y, x], pl[2][y, x], pl[3][y, x]])
    assert np.abs(got - want).max() <= 1.0 and got.max() > 3.0, (got, want, np.clip(c, 0, 1) * 255)


def test_displacement_scalar_tail_known_value(orc):
    """calcDisplacementMapping (src/Shader.cpp:447-475): the same perturbed normal as the bump shader AND the position moved along the
    ORIGINAL normal by kn * |texel| before the Blinn-Phong sum"""
    A, B, C = np.array([8.0, 8.0]), np.array([8.0, 28.5]), np.array([28.5, 8.0])
    rng = np.random.default_rng(12)
    tex = rng.integers(40, 216, (8, 8, 3)).astype(np.uint8)
    orc.texture_set(9, tex)
    nvec = np.array([-0.4, 0.2, -0.7])
    t = tri(tuple(A), tuple(B), tuple(C), z=30.0)
    t["nrm"][0] = [nvec] * 3
    t["uv"][0] = [[0.55, 0.20]] * 3
    lights = [((50.0, 35.0, -20.0), (7.0, 8.0, 9.0)), ((10.0, 2.0, 60.0), (2.0, 2.0, 2.0))]
    eye = (0.0, 0.0, 1.0)
    rc, pl, _ = orc.draw(frame(t, shader=abi.SHADER_DISPLACEMENT, tex=9, lights=lights, eye=eye))
    assert rc == 0
    x, y = 25, 9

    def texel(u, v):
        cu, cv = min(max(u, 0.0), 1.0), min(max(v, 0.0), 1.0)
        cx, cy = int(cu * 8), int(cv * 8)
        return np.zeros(3) if (cx >= 8 or cy >= 8) else tex[cy, cx].astype(float) / 255.0
    n = nvec / np.linalg.norm(nvec)
    s = np.sqrt(n[0] ** 2 + n[2] ** 2)
    tv = np.array([n[0] * n[1] / s, s, n[2] * n[1] / s])
    bv = np.cross(n, tv)
    kh, kn = 0.2, 0.1
    on = np.linalg.norm(texel(0.55, 0.20))
    dU = kh * kn * (np.linalg.norm(texel((0.55 + 1) / 8, 0.20)) - on)
    dV = kh * kn * (np.linalg.norm(texel(0.55, (0.20 + 1) / 8)) - on)
    ln = np.array([-dU, -dV, 1.0])
    nd = np.array([tv @ ln, bv @ ln, n @ ln])
    nd = nd / np.linalg.norm(nd)
    P = np.array([x, y, 30.0], float) + kn * n * on
    c = _blinn_phong64(P, nd, texel(0.55, 0.20), eye, lights)
    want = np.floor(np.clip(c, 0.0, 1.0) * 255.0)
    got = np.array([pl[1][y, x], pl[2][y, x], pl[3][y, x]])
    assert np.abs(got - want).max() <= 1.0 and got.max() > 3.0, (got, want, np.clip(c, 0, 1) * 255)


def test_texture_two_lights_known_value_v_class(orc):
    """an 8-wide-column pixel of the TEXTURE shader under two coloured lights: BlinnPhong<__m256> (include/shader/Shader.hpp:104-229) in
    closed form — attenuation 1 / |dxy| (2-D), normalised light and half vectors, x^150, kd = the texel picked by round-half-even of
    the clamped u * w (include/loader/TextureLoader.hpp:26-74) — summed, clamped, times 255, NOT truncated"""
    A, B, C = np.array([8.0, 8.0]), np.array([8.0, 28.5]), np.array([28.5, 8.0])
    rng = np.random.default_rng(13)
    tex = rng.integers(30, 226, (8, 8, 3)).astype(np.uint8)
    orc.texture_set(10, tex)
    nrm = np.array([[0.1, -0.2, -1.0], [0.3, 0.1, -0.9], [-0.2, 0.2, -0.95]])
    uv = np.array([[0.10, 0.20], [0.30, 0.85], [0.90, 0.35]])
    t = tri(tuple(A), tuple(B), tuple(C), z=(20.0, 35.0, 50.0))
    t["nrm"][0], t["uv"][0] = nrm, uv
    lights = [((45.0, 20.0, -15.0), (5.0, 7.0, 9.0)), ((3.0, 40.0, -40.0), (4.0, 3.0, 2.0))]
    eye = (0.0, 0.0, 1.0)
    rc, pl, st = orc.draw(frame(t, shader=abi.SHADER_TEXTURE, tex=10, lights=lights, eye=eye))
    assert rc == 0 and st["n_culled"] == 0
    for (x, y) in ((12, 9), (15, 14), (20, 10)):
        al, be, ga = _bary(np.array([float(x), float(y)]), A, B, C)
        assert min(al, be, ga) > 0
        z = al * 20.0 + be * 35.0 + ga * 50.0
        n = al * nrm[0] + be * nrm[1] + ga * nrm[2]
        u, v = al * uv[0] + be * uv[1] + ga * uv[2]
        col, row = int(np.rint(min(max(u * 8, 0.0), 7.0))), int(np.rint(min(max(v * 8, 0.0), 7.0)))
        kd = tex[row, col].astype(float) / 255.0
        c = _blinn_phong64(np.array([x, y, z], float), n, kd, eye, lights)
        want = np.clip(c, 0.0, 1.0) * 255.0
        got = np.array([pl[1][y, x], pl[2][y, x], pl[3][y, x]])
        assert np.abs(got - want).max() < 0.02, (x, y, got, want)
        assert got.max() > 1.0


def test_vertex_stage_of_a_whole_mesh_against_glm_formulas_in_binary64(orc):
    """Scene::loadTriangleStream (src/Scene.cpp:903-964) for the spot mesh, rotated, scaled and translated: the matrices of SURVEY.md
    Appendix B (glm::translate / rotate / scale, lookAtLH, perspectiveLH_NO with the raw 45 as radians, the NDC matrix with its x
    stretch) written out in numpy binary64, positions divided by w, depth remapped to [near, far], normals through the
    inverse-transpose of the model matrix with w = 1 and divided by w"""
    import scenes
    verts, faces = scenes.mesh(scenes.SPOT_OBJ)
    W, H, deg, tr, sc = 1920, 1080, 37.0, (-0.25, 0.05, 0.02), 0.3
    eye, center, up = np.array(scenes.EYE, float), np.zeros(3), np.array([0.0, 1.0, 0.0])
    got = scenes.mesh_stream(scenes.SPOT_OBJ, W, H, deg, tr, sc)

    def translate(t):
        m = np.eye(4)
        m[:3, 3] = t
        return m

    def rotate(a, axis):                                  # glm::rotate: Rodrigues
        n = np.asarray(axis, float) / np.linalg.norm(axis)
        c, s = np.cos(a), np.sin(a)
        K = np.array([[0, -n[2], n[1]], [n[2], 0, -n[0]], [-n[1], n[0], 0]])
        m = np.eye(4)
        m[:3, :3] = c * np.eye(3) + s * K + (1 - c) * np.outer(n, n)
        return m
    M = translate(tr) @ rotate(np.radians(deg), (0, 1, 0)) @ np.diag([sc, sc, sc, 1.0])
    f = (center - eye) / np.linalg.norm(center - eye)
    s_ = np.cross(up, f)
    s_ /= np.linalg.norm(s_)
    u_ = np.cross(f, s_)
    V = np.eye(4)
    V[0, :3], V[1, :3], V[2, :3] = s_, u_, f
    V[:3, 3] = [-s_ @ eye, -u_ @ eye, -f @ eye]
    n_, f_, aspect = 0.1, 100.0, W / H
    tg = np.tan(45.0 / 2.0)                               # the degrees fed as radians (src/Scene.cpp:293)
    P = np.zeros((4, 4))
    P[0, 0], P[1, 1], P[2, 2], P[2, 3], P[3, 2] = 1 / (aspect * tg), 1 / tg, (f_ + n_) / (f_ - n_), -2 * f_ * n_ / (f_ - n_), 1.0
    N = np.eye(4)
    N[0, 0], N[1, 1], N[0, 3], N[1, 3] = W / 2 * aspect, H / 2, W / 2, H / 2
    full = N @ P @ V @ M
    NM = np.linalg.inv(M).T
    p = np.c_[verts[:, :3].astype(float), np.ones(len(verts))] @ full.T
    p = p[:, :3] / p[:, 3:4]
    p[:, 2] = p[:, 2] * ((f_ - n_) / 2) + (f_ + n_) / 2
    nn = np.c_[verts[:, 3:6].astype(float), np.ones(len(verts))] @ NM.T
    nn = nn[:, :3] / nn[:, 3:4]
    want_pos, want_nrm = p[faces], nn[faces]
    assert got["pos"].shape == want_pos.shape == (len(faces), 3, 3)
    assert np.abs(got["pos"][:, :, :2] - want_pos[:, :, :2]).max() < 2e-2      # pixels (binary32 chain against binary64)
    assert np.abs(got["pos"][:, :, 2] - want_pos[:, :, 2]).max() < 2e-3        # depth in [0.1, 100]
    assert np.abs(got["nrm"] - want_nrm).max() < 1e-4 * np.abs(want_nrm).max()
    assert np.array_equal(got["uv"], verts[:, 6:8][faces])
    assert want_pos[:, :, 0].min() > 0 and want_pos[:, :, 0].max() < W and want_pos[:, :, 1].min() > 0 and want_pos[:, :, 1].max() < H   # on screen

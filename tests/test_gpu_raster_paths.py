"""-m gpu: the two rasterisers behind the C ABI against the CPU oracle.

k_raster resolves visibility ORDER-INDEPENDENTLY (one 64-bit depth | tie-break key per pixel, LDS min); what the keys
cannot express — a NaN depth that passes a scalar-tail test, a final depth of ±0, a band whose records did not fit the
pool — is handed to k_raster_slow, the reference's ordered triangle walk (src/Rasterizer.cpp:199-236).  Both must give
the oracle's framebuffer bit for bit; SRZ_ORDERED_RASTER forces the ordered one everywhere."""
import os

import numpy as np
import pytest
import torch

import scenes
from srz import abi
from test_oracle_kat import frame, tri

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


def same(gpu, ref, what):
    for p in range(4):
        bad = bits(gpu[p]) != bits(ref[p])
        assert not bad.any(), f"{what}: plane {p} differs at {int(bad.sum())} pixels, first {np.argwhere(bad)[:4].tolist()}"


def both_paths(ctx, orc, f_builder, planes_init=None, what=""):
    """draw through the order-independent rasteriser and through the ordered one; both must equal the oracle."""
    for extra in (0, abi.ORDERED_RASTER):
        f = f_builder(extra)
        clone = (lambda: None) if planes_init is None else (lambda: tuple(p.copy() for p in planes_init))
        rc, ref, rst = orc.draw(f, clone())
        assert rc == 0
        gpu, gst = ctx.draw(f, clone(), want_stats=True)
        assert gst == rst, (what, extra, gst, rst)
        same(gpu, ref, f"{what} flags+={extra}")


@pytest.mark.parametrize("shader", [abi.SHADER_TEXTURE, abi.SHADER_NORMAL])
def test_ordered_flag_equals_oracle_on_spot(ctx, orc, shader):
    both_paths(ctx, orc, lambda extra: scenes.config2(5, size=512, shader=shader, flags=abi.FUSED_CLEAR | extra), what="spot512")


def soup(seed, n, w, h, zs, big=False):
    rng = np.random.default_rng(seed)
    t = np.zeros(n, abi.TRI_DTYPE)
    c = rng.uniform(-8, [w + 8, h + 8], (n, 1, 2))
    r = rng.uniform(1, 70 if big else 24, (n, 1, 1))
    xy = c + rng.uniform(-1, 1, (n, 3, 2)) * r
    xy = np.round(xy * 4) / 4 if seed % 2 else xy  # half the seeds: quarter-pixel vertices → exact edge hits and ties
    t["pos"][:, :, :2] = xy
    t["pos"][:, :, 2] = rng.choice(zs, (n, 1)) if seed % 3 == 0 else rng.choice(zs, (n, 3))
    nn = rng.normal(size=(n, 3, 3))
    t["nrm"] = nn / np.linalg.norm(nn, axis=2, keepdims=True)
    t["uv"] = rng.uniform(0, 1, (n, 3, 2))
    return t


@pytest.mark.parametrize("seed", range(6))
def test_zero_and_signed_zero_depths(ctx, orc, seed):
    """depths drawn from {-0, +0, tiny, 1}: -0 == +0 as floats but not as keys — the ±0 pixels must take the ordered path"""
    zs = np.array([0.0, -0.0, 1e-30, -1e-30, 1.0], np.float32)
    t = soup(seed, 60, 96, 80, zs)
    both_paths(ctx, orc, lambda extra: frame(t, 96, 80, shader=abi.SHADER_NORMAL, flags=abi.FUSED_CLEAR | extra), what=f"zero-z seed {seed}")


@pytest.mark.parametrize("seed", range(4))
def test_negative_and_huge_depths(ctx, orc, seed):
    zs = np.array([-5.0, -1e30, 3e38, -3e38, 7.0, 7.0, 0.25], np.float32)
    t = soup(seed + 10, 80, 130, 70, zs, big=True)
    both_paths(ctx, orc, lambda extra: frame(t, 130, 70, shader=abi.SHADER_NORMAL, flags=abi.FUSED_CLEAR | extra), what=f"neg/huge seed {seed}")


@pytest.mark.parametrize("seed", range(4))
def test_incoming_planes_with_nan_inf_and_zero_depths(ctx, orc, seed):
    """accumulate mode: the incoming z plane holds NaN / ±inf / ±0 / ordinary depths.  A NaN in the buffer blocks every
    V fragment and admits every S fragment (src/Rasterizer.cpp:334,475)."""
    w, h = 100, 72
    rng = np.random.default_rng(100 + seed)
    z = rng.choice(np.array([np.nan, np.inf, -np.inf, 0.0, -0.0, 5.0, 50.0, 60.0], np.float32), (h, w)).astype(np.float32)
    init = (z, rng.uniform(0, 255, (h, w)).astype(np.float32), rng.uniform(0, 255, (h, w)).astype(np.float32),
            rng.uniform(0, 255, (h, w)).astype(np.float32))
    t = soup(seed + 20, 70, w, h, np.array([5.0, 50.0, 55.0, 60.0, 0.0], np.float32))
    both_paths(ctx, orc, lambda extra: frame(t, w, h, shader=abi.SHADER_NORMAL, flags=extra), planes_init=init, what=f"incoming seed {seed}")


def test_depth_ties_between_many_fragments(ctx, orc):
    """every triangle at the same depth: per pixel the LAST scalar-tail fragment wins if there is one, else the FIRST
    8-wide one — the tie-break half of the key"""
    t = soup(7, 120, 128, 96, np.array([42.0], np.float32), big=True)
    both_paths(ctx, orc, lambda extra: frame(t, 128, 96, shader=abi.SHADER_NORMAL, flags=abi.FUSED_CLEAR | extra), what="ties")


def test_unified_flag_both_paths(ctx, orc):
    t = soup(3, 90, 128, 96, np.array([1.0, 2.0, 3.0], np.float32), big=True)
    both_paths(ctx, orc, lambda extra: frame(t, 128, 96, shader=abi.SHADER_NORMAL, flags=abi.FUSED_CLEAR | abi.UNIFIED | extra), what="unified")


def test_pool_overflow_then_growth(ctx, orc, monkeypatch):
    """A few screen-filling triangles make far more (triangle, tile) pairs than the first guess of the list pool.  With the
    first-render sizing switched off (srz_set_option SRZ_OPT_POOL_LAZY, applied to sets created afterwards) the first render
    serves the bands that do not fit through the ordered rasteriser and the next one finds the pool grown.  Both must equal the
    oracle."""
    ctx.set_option(abi.OPT_POOL_LAZY, 1)
    try:  # (the shared ctx must get its default back whatever an assertion below does)
        w = h = 1024
        n = 24
        t = np.zeros(n, abi.TRI_DTYPE)
        rng = np.random.default_rng(5)
        for i in range(n):
            t["pos"][i] = [[-40 + 3 * i, -30, 10 + i % 5], [w + 50 - i, 10 + 2 * i, 12 + (i * 7) % 5], [200 + 5 * i, h + 60, 11 + (i * 3) % 7]]
        nn = rng.normal(size=(n, 3, 3))
        t["nrm"] = nn / np.linalg.norm(nn, axis=2, keepdims=True)
        # (eye behind the triangles' winding: none of them is culled — with test_oracle_kat.frame's default eye every one was, and
        # rounds 2-5 ran this test on an empty frame)
        f = frame(t, w, h, shader=abi.SHADER_NORMAL, eye=(0, 0, -1), flags=abi.FUSED_CLEAR)
        rc, ref, st = orc.draw(f)
        assert rc == 0 and st["n_culled"] == 0 and st["visible"] > 500000, st
        fs = ctx.frameset([f])
        out = torch.zeros(fs.out_shape, dtype=torch.float32, device="cuda")
        slow = []
        for it in range(3):
            out.fill_(-1.0)
            fs.render(out.data_ptr(), fs.out_bytes, abi.FUSED_CLEAR, torch.cuda.current_stream().cuda_stream)
            torch.cuda.synchronize()
            same(out[0].cpu().numpy(), ref, f"pool overflow, render {it}")
            slow.append(fs.debug_counters()["slow_tiles"])
        assert slow[0] > 0 and slow[-1] == 0, slow     # the first render's overflowing bands went the ordered way, the last one's none
        fs.close()
    finally:
        ctx.set_option(abi.OPT_POOL_LAZY, 0)
    with pytest.raises(Exception):
        ctx.set_option(99, 1)


def test_pool_is_sized_when_the_set_is_created(orc):
    """the same screen-filling triangles WITHOUT the lazy option: srz_frameset_create runs a binning pass of its own and sizes the
    pool, so the very first render (asynchronous like every other) is the fast path's and equals the oracle; a second context
    proves it for a sharded set too"""
    import srz
    w = h = 1024
    n = 24
    t = np.zeros(n, abi.TRI_DTYPE)
    rng = np.random.default_rng(6)
    for i in range(n):
        t["pos"][i] = [[-40 + 3 * i, -30, 10 + i % 5], [w + 50 - i, 10 + 2 * i, 12 + (i * 7) % 5], [200 + 5 * i, h + 60, 11 + (i * 3) % 7]]
    nn = rng.normal(size=(n, 3, 3))
    t["nrm"] = nn / np.linalg.norm(nn, axis=2, keepdims=True)
    f = frame(t, w, h, shader=abi.SHADER_NORMAL, eye=(0, 0, -1), flags=abi.FUSED_CLEAR)
    rc, ref, st = orc.draw(f)
    assert rc == 0 and st["n_culled"] == 0 and st["visible"] > 500000, st
    for (rank, world) in ((0, 1), (1, 2)):
        c = srz.Context(0, rank, world)
        fs = c.frameset([f, f])
        out = torch.full(fs.out_shape, -1.0, dtype=torch.float32, device="cuda")
        fs.render(out.data_ptr(), fs.out_bytes, abi.FUSED_CLEAR, torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        # the FIRST render took the fast path everywhere: no band overflowed the pool creation sized (the ordered rasteriser would
        # give the same pixels, so the pixels alone do not prove it)
        dc = fs.debug_counters()
        assert dc["slow_tiles"] == 0 and dc["pool_sub_cap"] >= dc["pool_demand"] > 0, dc
        got = out.cpu().numpy()
        if world == 1:
            same(got[0], ref, "sized at creation, first render")
            same(got[1], ref, "sized at creation, first render, frame 1")
        else:
            from srz import parallel
            for (lb, b, r0, r1) in parallel.band_rows(h, rank, world):
                assert np.array_equal(got[0][:, lb * 32: lb * 32 + (r1 - r0)].view(np.uint32), np.stack(ref)[:, r0:r1].view(np.uint32)), (rank, b)
        fs.close(), c.close()


def test_many_small_frames_fill_every_subpool(ctx, orc):
    """enough frames that the pool is split into several sub-pools; every frame must still come out right"""
    frames = [scenes.config2(i, size=256) for i in range(40)]
    fs = ctx.frameset(frames)
    out = torch.zeros(fs.out_shape, dtype=torch.float32, device="cuda")
    for it in range(2):
        fs.render(out.data_ptr(), fs.out_bytes, abi.FUSED_CLEAR, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    got = out.cpu().numpy()
    for i in (0, 9, 17, 39):
        rc, ref, _ = orc.draw(frames[i])
        same(got[i], ref, f"frame {i}")
    fs.close()


def big_tris(n, w, h, seed, tall):
    """n triangles about `tall` pixels high at random places (depths distinct per triangle)"""
    rng = np.random.default_rng(seed)
    t = np.zeros(n, abi.TRI_DTYPE)
    cx, cy = rng.uniform(0, w, n), rng.uniform(0, h, n)
    t["pos"][:, 0, :2] = np.stack([cx - 9, cy - tall / 2], 1)
    t["pos"][:, 2, :2] = np.stack([cx + 11, cy - tall / 2 + 3], 1)  # (this winding faces the eye at (0, 0, 1))
    t["pos"][:, 1, :2] = np.stack([cx + 2, cy + tall / 2], 1)
    t["pos"][:, :, 2] = rng.uniform(5, 50, (n, 1))
    nn = rng.normal(size=(n, 3, 3))
    t["nrm"] = nn / np.linalg.norm(nn, axis=2, keepdims=True)
    return t


def test_first_render_sizes_the_pool(ctx, orc):
    """the same screen-filling triangles WITHOUT the lazy switch: a one-shot set must come out of the order-independent
    rasteriser on its first (and only) render — no tile may be left to the ordered one for want of pool space"""
    w = h = 1024
    n = 24
    t = np.zeros(n, abi.TRI_DTYPE)
    for i in range(n):
        t["pos"][i] = [[-40 + 3 * i, -30, 10 + i % 5], [w + 50 - i, 10 + 2 * i, 12 + (i * 7) % 5], [200 + 5 * i, h + 60, 11 + (i * 3) % 7]]
    t["nrm"][:] = [0, 0, -1]
    f = frame(t, w, h, shader=abi.SHADER_NORMAL, flags=abi.FUSED_CLEAR)
    rc, ref, _ = orc.draw(f)
    planes, _ = ctx.draw_batch([f])
    same(planes[0], ref, "one-shot draw_batch")


@pytest.mark.parametrize("tall,n", [(150, 700), (2300, 40)])
def test_groups_with_huge_triangles_take_the_raw_walk(ctx, orc, tall, n):
    """a 512-triangle group whose triangles reach more bands than its entry region holds (700 triangles x 5-6 bands), and
    triangles taller than 64 bands: k_setup marks the group DESC_RAW and k_bin's band workgroups walk its bounding boxes"""
    w, h = 160, 2400
    t = big_tris(n, w, h, 11, tall)
    both_paths(ctx, orc, lambda extra: frame(t, w, h, shader=abi.SHADER_NORMAL, flags=abi.FUSED_CLEAR | extra), what=f"raw groups tall={tall}")


def test_band_with_more_pairs_than_the_lds_stage(ctx, orc):
    """6000 small triangles in ONE 32-row band of a wide frame: the band's index list (> 4096 entries) does not fit k_bin's LDS
    stage and is stored straight into the pool"""
    w, h = 2048, 64
    rng = np.random.default_rng(3)
    n = 6000
    t = np.zeros(n, abi.TRI_DTYPE)
    cx, cy = rng.uniform(4, w - 4, n), rng.uniform(3, 28, n)
    t["pos"][:, 0, :2] = np.stack([cx - 3, cy - 2], 1)
    t["pos"][:, 2, :2] = np.stack([cx + 4, cy - 1], 1)
    t["pos"][:, 1, :2] = np.stack([cx, cy + 3], 1)
    t["pos"][:, :, 2] = rng.uniform(5, 50, (n, 1))
    t["nrm"][:] = [0, 0, -1]
    both_paths(ctx, orc, lambda extra: frame(t, w, h, shader=abi.SHADER_NORMAL, flags=abi.FUSED_CLEAR | extra), what="stage overflow")


def adversarial_tris(seed, n, w, h):
    """the shapes the tightened rectangles (k_raster's slab clips, bucket_group's band clips) must stay conservative for:
    needles, slivers, huge and far-off-screen vertices (past the 2^20 guard too), sub-pixel triangles around pixel centres,
    edges exactly through pixel centres and along tile borders, ordinary large triangles — in both windings"""
    rng = np.random.default_rng(seed)
    t = np.zeros(n, abi.TRI_DTYPE)
    kind = rng.integers(0, 9, n)
    c = rng.uniform([0, 0], [w, h], (n, 2))
    ang = rng.uniform(0, 2 * np.pi, n)
    d = np.stack([np.cos(ang), np.sin(ang)], 1)
    nrm = np.stack([-d[:, 1], d[:, 0]], 1)
    xy = np.zeros((n, 3, 2))
    L = rng.uniform(100, 600, n)[:, None]
    # 0 needles: a very short base, the apex far away
    k = kind == 0
    base = (10.0 ** rng.uniform(-3, 0, n))[:, None]
    xy[k] = np.stack([c, c + nrm * base, c + d * L], 1)[k]
    # 1 slivers: a long edge, a height of 1e-4 .. 0.5 pixels
    k = kind == 1
    hgt = (10.0 ** rng.uniform(-4, -0.3, n))[:, None]
    xy[k] = np.stack([c, c + d * L, c + d * L * rng.uniform(0, 1, (n, 1)) + nrm * hgt], 1)[k]
    # 2 huge: vertices 1e3 .. 1e6 pixels out;  3 past the guard: one vertex 2e6 .. 1e8 out
    k = kind == 2
    xy[k] = (c[:, None, :] + rng.normal(size=(n, 3, 2)) * (10.0 ** rng.uniform(3, 6, (n, 1, 1))))[k]
    k = kind == 3
    far = c[:, None, :] + rng.normal(size=(n, 3, 2)) * 300.0
    far[:, 0] += d * (10.0 ** rng.uniform(6.3, 8, n))[:, None]
    xy[k] = far[k]
    # 4 sub-pixel triangles around pixel centres
    k = kind == 4
    xy[k] = (np.round(c)[:, None, :] + rng.uniform(-1, 1, (n, 3, 2)) * (10.0 ** rng.uniform(-3, 0, (n, 1, 1))))[k]
    # 5 integer / half-integer / tile-border vertices: edges through pixel centres, exact zeros of the edge functions
    k = kind == 5
    grid = rng.choice([1.0, 0.5, 32.0], (n, 1, 1))
    xy[k] = (np.round((c[:, None, :] + rng.uniform(-1, 1, (n, 3, 2)) * rng.uniform(2, 200, (n, 1, 1))) / grid) * grid)[k]
    # 6 ordinary large triangles
    k = kind == 6
    xy[k] = (c[:, None, :] + rng.uniform(-1, 1, (n, 3, 2)) * rng.uniform(30, 400, (n, 1, 1)))[k]
    # 7 tiny triangles (extent 2^-12 .. 2^-4 pixels) at coordinates < 64 around pixel centres, near the sliver limit of the guard:
    #   below 2^-5 the tightened rectangles must fall back to the plain box (tight_margin), above it the margin must hold
    k = kind == 7
    Dt = (2.0 ** rng.uniform(-12, -4, n))[:, None]
    c7 = np.round(rng.uniform([0, 0], [64, 64], (n, 2))) + rng.uniform(-1, 1, (n, 2)) * Dt
    xy[k] = np.stack([c7, c7 + d * Dt, c7 + d * Dt * rng.uniform(0, 1, (n, 1)) + nrm * Dt * (2.0 ** rng.uniform(-7.5, 0, n))[:, None]], 1)[k]
    # 8 small or ulp-sized triangles 1e4 .. 1e7 pixels off screen on ONE axis: the clamped box is an edge column / row at a
    #   distance from the triangle that has nothing to do with its extent (plain box there)
    k = kind == 8
    off = np.zeros((n, 2))
    off[np.arange(n), rng.integers(0, 2, n)] = rng.choice([-1.0, 1.0], n) * 10.0 ** rng.uniform(4, 7, n)
    D8 = (10.0 ** rng.uniform(-3, 2.5, n))[:, None, None]
    xy[k] = ((c + off)[:, None, :] + rng.uniform(-1, 1, (n, 3, 2)) * D8)[k]
    flip = rng.random(n) < 0.5
    xy[flip] = xy[flip][:, ::-1]
    t["pos"][:, :, :2] = xy
    # (the screen-filling kinds lie behind the others, so that every small shape decides pixels of the final image)
    t["pos"][:, :, 2] = np.where((kind == 2) | (kind == 3), rng.uniform(60, 80, n), rng.uniform(2, 60, n))[:, None] + rng.uniform(-1, 1, (n, 3))
    nn = rng.normal(size=(n, 3, 3))
    t["nrm"] = nn / np.linalg.norm(nn, axis=2, keepdims=True)
    t["uv"] = rng.uniform(0, 1, (n, 3, 2))
    return t


@pytest.mark.parametrize("seed", range(int(__import__("os").environ.get("SRZ_ADVERSARIAL_SEEDS", "6"))))
def test_tight_rectangles_stay_conservative_on_adversarial_shapes(ctx, orc, seed):
    """k_raster walks, and k_bin lists, the rectangle of a triangle tightened by slab clips with a margin (tight_margin): every
    pixel the reference's bounding-box walk would set must still be tested.  Against the oracle (plain bounding boxes) and against
    the ordered rasteriser (plain bounding boxes too), on shapes built to break a careless clip."""
    w, h = (640, 416) if seed % 2 == 0 else (333, 517)
    t = adversarial_tris(40 + seed, 800, w, h)
    sh = [abi.SHADER_NORMAL, abi.SHADER_PHONG, abi.SHADER_TEXTURE][seed % 3]
    tex = scenes.TEX_SPOT if sh == abi.SHADER_TEXTURE else -1
    fl = abi.FUSED_CLEAR | (abi.UNIFIED if seed == 5 else 0)
    lights = [((100.0, 100.0, -50.0), (300.0, 300.0, 300.0)), ((400.0, 50.0, 80.0), (200.0, 200.0, 200.0))]
    both_paths(ctx, orc, lambda extra: frame(t, w, h, shader=sh, tex=tex, lights=lights, flags=fl | extra), what=f"adversarial shapes {seed}")


def test_tight_rectangles_in_the_throughput_build(ctx, orc):
    """the same shapes as a BATCH: 18 frames of 20 x 13 tiles = 4680 tiles — above the 4096 of the four-waves-per-tile build, so
    this is k_raster<1> (which clears the tiles no bbox reaches itself: the side-stream clear starts at 8192 tiles) and the frameset
    path; every frame against the oracle"""
    w, h = 640, 416
    lights = [((100.0, 100.0, -50.0), (300.0, 300.0, 300.0)), ((400.0, 50.0, 80.0), (200.0, 200.0, 200.0))]
    shaders = [abi.SHADER_NORMAL, abi.SHADER_PHONG, abi.SHADER_TEXTURE]
    frames = [frame(adversarial_tris(900 + i, 350, w, h), w, h, shader=shaders[i % 3], tex=scenes.TEX_SPOT if i % 3 == 2 else -1,
                    lights=lights, flags=abi.FUSED_CLEAR) for i in range(18)]
    fs = ctx.frameset(frames)
    out = torch.zeros(fs.out_shape, dtype=torch.float32, device="cuda")
    for _ in range(2):  # (the second render runs with the pool sized by the first)
        fs.render(out.data_ptr(), fs.out_bytes, 0, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    got = out.cpu().numpy()
    for i, f in enumerate(frames):
        rc, ref, _ = orc.draw(f)
        assert rc == 0
        same(got[i], ref, f"batch frame {i}")


@pytest.mark.parametrize("rank,world,grid", [(0, 1, None), (1, 3, None), (0, 1, "37"), (0, 1, "256"), (2, 3, "160")])
def test_side_stream_clear_odd_width_ragged_height(orc, monkeypatch, rank, world, grid):
    """k_clear on its side stream (sets of >= 8192 tiles) on frames whose width is no multiple of 4 (scalar stores, the last quad cut
    by the frame's edge) and whose height is no multiple of 32 (a ragged last band), most tiles untouched, the buffer poisoned with
    NaN before every render — unsharded and as a rank of 3 (bands dealt by band_of; the shard's rows against the oracle's); with the
    grid measured by the set (rendered often enough for that) and with fixed grids (SRZ_CLEAR_WGS, read when the ctx is created:
    the measurement does not try a larger grid once one is clearly behind)"""
    import srz
    if grid:
        monkeypatch.setenv("SRZ_CLEAR_WGS", grid)
    from srz import parallel
    w, h = 333, 301   # 11 x 10 tiles
    n_frames = 84 if world == 1 else 252
    uniq = [frame(adversarial_tris(4000 + i, 25, w, h), w, h, shader=abi.SHADER_NORMAL, flags=abi.FUSED_CLEAR) for i in range(12)]
    refs = [np.stack(orc.draw(f)[1]) for f in uniq]
    c = srz.Context(0, rank, world)
    fs = c.frameset([uniq[i % 12] for i in range(n_frames)])
    out = torch.empty(fs.out_shape, dtype=torch.float32, device="cuda")
    rows = parallel.band_rows(h, rank, world)
    s = torch.cuda.current_stream().cuda_stream
    for it in range(26):
        out.fill_(float("nan"))
        fs.render(out.data_ptr(), fs.out_bytes, abi.FUSED_CLEAR, s)
        torch.cuda.synchronize()
        if it in (0, 7, 10, 13, 25):   # (first render, one render of each candidate's first block, the measured grid)
            got = out.cpu().numpy()
            for i in (0, 5, 11, n_frames - 1):
                ref = refs[i % 12]
                for (lb, b, r0, r1) in rows:
                    a_ = got[i][:, lb * 32: lb * 32 + (r1 - r0)].view(np.uint32)
                    assert np.array_equal(a_, ref[:, r0:r1].view(np.uint32)), (it, i, b)
    dc = fs.debug_counters()
    assert (dc["clear_wgs"] == int(grid)) if grid else (dc["clear_tuned"] == 1 and dc["clear_wgs"] in (96, 128, 256)), dc
    fs.close(), c.close()


@pytest.mark.parametrize("seed", range(6))
def test_side_stream_clear_random_shapes(orc, seed):
    """the same for seeded random frame shapes (33 .. 700 pixels each way, whatever that makes of quads, tile columns and last bands),
    as many frames as make 8192 tiles, a random rank of a random world, the grid measured or fixed at random"""
    import srz
    from srz import parallel
    rng = np.random.default_rng(7000 + seed)
    w, h = int(rng.integers(33, 701)), int(rng.integers(33, 701))
    world = int(rng.integers(1, 5))
    rank = int(rng.integers(0, world))
    tiles = ((w + 31) // 32) * len(parallel.band_rows(h, rank, world))
    n_frames = -(-8200 // max(tiles, 1))
    uniq = [frame(adversarial_tris(5000 + 10 * seed + i, int(rng.integers(1, 30)), w, h), w, h, shader=abi.SHADER_NORMAL, flags=abi.FUSED_CLEAR)
            for i in range(6)]
    refs = [np.stack(orc.draw(f)[1]) for f in uniq]
    grid = [None, "64", "200", None, "17", None][seed]
    if grid:
        os.environ["SRZ_CLEAR_WGS"] = grid
    try:
        c = srz.Context(0, rank, world)
    finally:
        os.environ.pop("SRZ_CLEAR_WGS", None)
    fs = c.frameset([uniq[i % 6] for i in range(n_frames)])
    out = torch.empty(fs.out_shape, dtype=torch.float32, device="cuda")
    rows = parallel.band_rows(h, rank, world)
    s = torch.cuda.current_stream().cuda_stream
    for it in range(26 if grid is None else 2):
        out.fill_(float("nan"))
        fs.render(out.data_ptr(), fs.out_bytes, abi.FUSED_CLEAR, s)
        torch.cuda.synchronize()
        if it in (0, 1, 7, 10, 13, 25):
            got = out.cpu().numpy()
            for i in sorted({0, 3, 5, n_frames // 2, n_frames - 1}):
                ref = refs[i % 6]
                for (lb, b, r0, r1) in rows:
                    a_ = got[i][:, lb * 32: lb * 32 + (r1 - r0)].view(np.uint32)
                    assert np.array_equal(a_, ref[:, r0:r1].view(np.uint32)), (seed, w, h, rank, world, it, i, b)
    dc = fs.debug_counters()   # (a measured grid proves the set is large enough for the side-stream clear: smaller sets never measure)
    assert (dc["clear_wgs"] == int(grid)) if grid else (dc["clear_tuned"] == 1), (dc, w, h, rank, world, n_frames)
    fs.close(), c.close()


def test_owner_ids_by_triangle_index(orc, monkeypatch):
    """frames of 2^22 - 1 triangles or more, or of more than 1024 batches, are not FD_PACKED: their tile lists hold plain triangle
    indices, the owner ids are 32-bit indices and k_shade gathers the triangles per pixel instead of staging them.  SRZ_NO_PACKED
    (read when the ctx is created) forces that form: same planes bit for bit, both rasterisers, batch and single frame"""
    import srz
    monkeypatch.setenv("SRZ_NO_PACKED", "1")
    c = srz.Context(0)
    c.texture_upload(scenes.TEX_SPOT, scenes.spot_texture())
    try:
        f = scenes.config2(7, size=512)
        rc, ref, _ = orc.draw(f)
        assert rc == 0
        fs = c.frameset([scenes.config2(i, size=512) for i in (7, 8, 9)] * 7)   # 21 frames x 256 tiles: the throughput build
        out = torch.zeros(fs.out_shape, dtype=torch.float32, device="cuda")
        for extra in (0, abi.ORDERED_RASTER):
            out.fill_(-2.0)
            fs.render(out.data_ptr(), fs.out_bytes, abi.FUSED_CLEAR | extra, torch.cuda.current_stream().cuda_stream)
            torch.cuda.synchronize()
            same(out[0].cpu().numpy(), ref, f"ids by index, batch, flags {extra}")
            same(out[19].cpu().numpy(), orc.draw(scenes.config2(8, size=512))[1], "ids by index, frame 19")
        fs.close()
        fs1 = c.frameset([f])                                                    # one frame: four waves per tile
        o1 = torch.zeros(fs1.out_shape, dtype=torch.float32, device="cuda")
        fs1.render(o1.data_ptr(), fs1.out_bytes, abi.FUSED_CLEAR, torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        same(o1[0].cpu().numpy(), ref, "ids by index, single frame")
        fs1.close()
    finally:
        c.close()


def test_long_tile_lists_and_many_batches(ctx, orc):
    """tiles whose triangle list is longer than k_shade's LDS stage (96 entries: position -> index -> gather) and longer than the
    tie-break's position field (512: ids by index), and a frame of 70 batches (more than the 16 staged shader descriptors): 600
    small triangles stacked on a few tiles, every batch with its own shader type (half of the triangles survive the backface test)"""
    rng = np.random.default_rng(77)
    w, h = 160, 96
    for n, nb in ((400, 3), (1800, 70)):
        t = np.zeros(n, abi.TRI_DTYPE)
        c = rng.uniform([20, 20], [60, 60], (n, 2))
        t["pos"][:, :, :2] = c[:, None, :] + rng.uniform(-14, 14, (n, 3, 2))
        t["pos"][:, :, 2] = rng.uniform(2, 60, (n, 1)) + rng.uniform(-1, 1, (n, 3))
        nn = rng.normal(size=(n, 3, 3))
        t["nrm"] = nn / np.linalg.norm(nn, axis=2, keepdims=True)
        t["uv"] = rng.uniform(0, 1, (n, 3, 2))
        cuts = np.linspace(0, n, nb + 1).astype(int)
        shaders = [abi.SHADER_NORMAL, abi.SHADER_PHONG, abi.SHADER_TEXTURE]
        batches = [(shaders[b % 3], scenes.TEX_SPOT if b % 3 == 2 else -1, t[cuts[b]:cuts[b + 1]]) for b in range(nb)]
        lights = [((100.0, 100.0, -50.0), (300.0, 300.0, 300.0)), ((40.0, 50.0, 80.0), (200.0, 200.0, 200.0))]
        both_paths(ctx, orc, lambda extra: abi.Frame(w, h, (0.0, 0.0, 0.9), lights, batches, abi.FUSED_CLEAR | extra), what=f"long lists {n}/{nb}")

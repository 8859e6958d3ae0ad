"""-m gpu: the throughput path (frames resident in HBM, one launch set per batch) and the band sharding, all through
the C ABI.  A single GPU can play every rank of an N-GPU job, so the sharded outputs are checked here row by row."""
import numpy as np
import pytest
import torch

import scenes
from srz import abi, parallel

pytestmark = pytest.mark.gpu


def bits(a):
    return np.ascontiguousarray(a).view(np.uint32)


@pytest.fixture(scope="module")
def frames():
    return [scenes.config2(i, size=512) for i in (0, 3, 7, 12, 30)]


def render(ctx, frames, flags=abi.FUSED_CLEAR, out=None):
    fs = ctx.frameset(frames)
    if out is None:
        out = torch.zeros(fs.out_shape, dtype=torch.float32, device="cuda")
    fs.render(out.data_ptr(), fs.out_bytes, flags, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    return fs, out


def test_frameset_equals_per_frame_draw_and_oracle(orc, frames):
    import srz
    ctx = srz.Context(0)
    ctx.texture_upload(0, scenes.spot_texture())
    fs, out = render(ctx, frames)
    got = out.cpu().numpy()
    tot = {}
    for i, f in enumerate(frames):
        rc, ref, st = orc.draw(f)
        for p in range(4):
            assert np.array_equal(bits(got[i, p]), bits(ref[p])), (i, p)
        for k, v in st.items():
            tot[k] = tot.get(k, 0) + v
    assert fs.stats() == tot
    # algorithmic bytes = 16*W*H + 96*N_tri + 24*N_lights + min(3*texW*texH, 3*textured pixels), per frame
    expect = sum(16 * 512 * 512 + 96 * f.n_tris + 24 * 2 for f in frames) + min(5 * 3 * 1024 * 1024, 3 * tot["visible_textured"])
    assert fs.algorithmic_bytes() == expect
    ctx.close()


def test_both_triangle_stream_layouts_give_the_same_planes(frames):
    """a frameset keeps a dense copy of the positions for k_setup / k_raster (pos_stride 9); srz_draw re-uploads one frame's 96-byte
    records per call and the same kernels read the positions out of them (pos_stride 24).  Same frame, both ways, bit for bit —
    twice, so that the second srz_draw goes through the re-upload of an existing set"""
    import srz
    ctx = srz.Context(0)
    ctx.texture_upload(0, scenes.spot_texture())
    _, out = render(ctx, frames)
    got = out.cpu().numpy()
    for rep in range(2):
        for i, f in enumerate(frames):
            planes, _ = ctx.draw(f)
            for p in range(4):
                assert np.array_equal(bits(got[i, p]), bits(planes[p])), (rep, i, p)
    ctx.close()


def test_kernel_timing_api(frames):
    import srz
    ctx = srz.Context(0)
    ctx.texture_upload(0, scenes.spot_texture())
    fs = ctx.frameset(frames)
    out = torch.empty(fs.out_shape, dtype=torch.float32, device="cuda")
    ctx.set_kernel_timing(True)
    for _ in range(3):
        fs.render(out.data_ptr(), fs.out_bytes, abi.FUSED_CLEAR, torch.cuda.current_stream().cuda_stream)
    kt = ctx.kernel_time_ms()
    assert kt["launches"] == 3 and kt["total_ms"] > 0 and kt["raster_ms"] > 0
    assert abs(kt["bin_ms"] + kt["raster_ms"] + kt["shade_ms"] - kt["total_ms"]) < 0.2 * kt["total_ms"] + 0.05
    ctx.close()


def test_frameset_argument_checks(frames):
    import srz
    ctx = srz.Context(0)
    with pytest.raises(srz.SrzError):
        ctx.frameset([scenes.config2(0, size=512), scenes.config2(0, size=256)])     # mixed sizes
    fs = ctx.frameset(frames[:1])
    out = torch.empty(fs.out_shape, dtype=torch.float32, device="cuda")
    with pytest.raises(srz.SrzError) as e:                                           # texture slot 0 never uploaded
        fs.render(out.data_ptr(), fs.out_bytes)
    assert e.value.code == abi.SRZ_E_TEXTURE
    ctx.texture_upload(0, scenes.spot_texture())
    with pytest.raises(srz.SrzError):
        fs.render(out.data_ptr(), fs.out_bytes - 4)                                  # buffer too small
    ctx.close()


@pytest.mark.parametrize("world,w,h", [(2, 512, 512), (3, 320, 200), (8, 256, 1080)])
def test_band_sharding_every_rank_on_one_gpu(orc, world, w, h):
    """Each rank's shard holds exactly its bands of the oracle frame; all-gather layout + de-interleave rebuild it."""
    import srz
    fr = [scenes.config2(i, size=w) if w == h else scenes.config3(i, w, h) for i in (2, 9)]
    refs = [np.stack(orc.draw(f)[1]) for f in fr]
    shards = []
    for rank in range(world):
        ctx = srz.Context(0, rank, world)
        ctx.texture_upload(0, scenes.spot_texture())
        fs, out = render(ctx, fr)
        lay = parallel.shard_layout(h, rank, world)
        assert fs.local_rows == lay["local_rows"] and tuple(out.shape) == (2, 4, lay["local_rows"], w)
        host = out.cpu().numpy()
        for (lb, band, r0, r1) in parallel.band_rows(h, rank, world):
            for i in range(2):
                assert np.array_equal(bits(host[i, :, lb * 32: lb * 32 + (r1 - r0)]), bits(refs[i][:, r0:r1])), (rank, band)
        shards.append(out.cpu())
        ctx.close()
    full = parallel.deinterleave(torch.stack(shards), world).numpy()[:, :, :h]
    for i in range(2):
        assert np.array_equal(bits(full[i]), bits(refs[i]))


def test_full_size_properties_config2_batch(orc):
    """At BASELINE's full size with a 36-frame batch (large enough for the side-stream clear): determinism, frame
    independence, a checksum of checksums, and three of the frames against the oracle bit for bit."""
    import srz
    ctx = srz.Context(0)
    ctx.texture_upload(0, scenes.spot_texture())
    fr = [scenes.config2(i) for i in range(36)]
    fs, out = render(ctx, fr)
    a = out.clone()
    fs.render(out.data_ptr(), fs.out_bytes, abi.FUSED_CLEAR, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    assert torch.equal(a.view(torch.int32), out.view(torch.int32))                  # deterministic
    fs2, out2 = render(ctx, list(reversed(fr)))
    assert torch.equal(out2.flip(0).view(torch.int32), a.view(torch.int32))         # frames are independent of batch order
    cov = torch.isfinite(a[:, 0]).sum(dim=(1, 2)).cpu().numpy()
    st = fs.stats()
    assert int(cov.sum()) == st["visible"] and (cov > 90_000).all() and (cov < 200_000).all()
    got = a.cpu().numpy()
    for i in (0, 17, 35):
        rc, ref, _ = orc.draw(fr[i])
        for p in range(4):
            assert np.array_equal(bits(got[i, p]), bits(ref[p])), (i, p)
    ctx.close()


def test_device_resolve8_equals_display_resolve(orc, frames):
    """RenderingPipeline::display's merge + convertTo(CV_8UC3) on the device vs the oracle's restatement."""
    import srz
    ctx = srz.Context(0)
    ctx.texture_upload(0, scenes.spot_texture())
    fs, out = render(ctx, frames)
    # salt a few values to exercise rounding ties, saturation and NaN
    out[0, 1, 0, :8] = torch.tensor([0.5, 1.5, 2.5, 254.5, 255.5, 300.0, -3.0, float("nan")], device="cuda")
    bgr = torch.zeros((fs.n_frames, fs.local_rows, fs.width, 3), dtype=torch.uint8, device="cuda")
    fs.resolve8(out.data_ptr(), bgr.data_ptr(), bgr.numel(), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    host = out.cpu().numpy()
    got = bgr.cpu().numpy()
    for i in range(fs.n_frames):
        assert np.array_equal(got[i], orc.resolve8(tuple(host[i])))
    assert got[0, 0, :8, 0].tolist() == [0, 2, 2, 254, 255, 255, 0, 0]
    ctx.close()


def _random_frame(rng, w, h, n_tris, flags):
    """Random soup: mostly small triangles, some large / off-screen / sliver ones, random normals, uvs beyond [0,1],
    every shader, 0-3 lights — the shapes the tile masks, the per-frame work lists and the FastMath paths must survive."""
    def tris(n):
        t = np.zeros(n, abi.TRI_DTYPE)
        c = rng.uniform([-0.1 * w, -0.1 * h], [1.1 * w, 1.1 * h], (n, 1, 2))
        size = np.where(rng.random((n, 1, 1)) < 0.85, rng.uniform(1, 24, (n, 1, 1)), rng.uniform(24, 1.5 * max(w, h), (n, 1, 1)))
        xy = c + rng.uniform(-1, 1, (n, 3, 2)) * size
        snap = rng.random((n, 1, 1)) < 0.3          # vertices exactly on pixel corners: on-edge samples, exact zeros
        xy = np.where(snap, np.round(xy), xy)
        t["pos"][:, :, :2] = xy
        t["pos"][:, :, 2] = rng.uniform(1, 90, (n, 1)) + rng.uniform(-0.5, 0.5, (n, 3))
        t["nrm"] = rng.normal(0, 1, (n, 3, 3)) * rng.choice([1.0, 1e-3, 50.0], (n, 1, 1))
        t["uv"] = rng.uniform(-0.2, 1.2, (n, 3, 2))
        return t
    shaders = [abi.SHADER_NORMAL, abi.SHADER_TEXTURE, abi.SHADER_PHONG, abi.SHADER_BUMP, abi.SHADER_DISPLACEMENT]
    nb = int(rng.integers(1, 4))
    batches = []
    for b in range(nb):
        sh = shaders[int(rng.integers(0, len(shaders)))]
        batches.append((sh, 0 if sh in (abi.SHADER_TEXTURE, abi.SHADER_BUMP, abi.SHADER_DISPLACEMENT) else -1, tris(max(1, n_tris // nb))))
    nl = int(rng.integers(0, 4))
    lights = np.concatenate([rng.uniform([0, 0, -50], [w, h, 120], (nl, 1, 3)), rng.uniform(0, 400, (nl, 1, 3))], 1).astype(np.float32)
    return abi.Frame(w, h, (0.0, 0.0, float(rng.uniform(0.5, 2.0))), lights, batches, flags, p=float(rng.choice([150.0, 8.0, 2.5])))


import os


@pytest.mark.parametrize("seed", range(int(os.environ.get("SRZ_FUZZ_SEEDS", "8"))))
def test_fuzz_random_frames_bit_identical_to_oracle(orc, seed):
    """Seeded random batches (different content, batch structure, light count and exponent per frame) against the oracle."""
    import srz
    rng = np.random.default_rng(1000 + seed)
    w, h = [(64, 64), (200, 120), (97, 131), (256, 96), (33, 290), (128, 128), (320, 200), (70, 70)][seed % 8]
    flags = abi.FUSED_CLEAR | (abi.UNIFIED if seed % 3 == 2 else 0)
    frames = [_random_frame(rng, w, h, int(rng.integers(1, 400)), flags) for _ in range(int(rng.integers(2, 12)))]
    ctx = srz.Context(0)
    ctx.texture_upload(0, scenes.spot_texture())
    fs, out = render(ctx, frames, flags=0)           # flags come from the frames
    got = out.cpu().numpy()
    tot = {}
    for i, f in enumerate(frames):
        rc, ref, st = orc.draw(f)
        assert rc == 0
        for p in range(4):
            same = bits(got[i, p]) == bits(ref[p])
            assert same.all(), f"seed {seed} frame {i} plane {p}: {int((~same).sum())} words differ, first at {np.argwhere(~same)[0]}"
        for k, v in st.items():
            tot[k] = tot.get(k, 0) + v
    assert fs.stats() == tot
    ctx.close()


def test_draw_batch_host_buffers_equal_per_frame_draw(orc):
    """srz_draw_batch (host planes in/out, one launch set) = srz_draw frame by frame = the oracle; one frame of the batch
    accumulates into caller-provided planes (no clear), the others are fused-clear."""
    import srz
    ctx = srz.Context(0)
    ctx.texture_upload(0, scenes.spot_texture())
    fr = [scenes.config2(i, size=256) for i in (0, 5, 9)]
    fr[1] = scenes.config2(5, size=256, flags=0)                      # accumulate: starts from the given planes
    planes = np.zeros((3, 4, 256, 256), np.float32)
    planes[:, 0] = np.inf
    planes[1, 0, 100:140] = 1.0                                        # a near "wall" across 40 rows: occludes the spot there
    planes[1, 1:4, 100:140] = 77.0
    init1 = tuple(planes[1, p].copy() for p in range(4))
    got, st = ctx.draw_batch(fr, planes, want_stats=True)
    tot = {}
    for i, f in enumerate(fr):
        rc, ref, s1 = orc.draw(f, init1 if i == 1 else None)
        assert rc == 0
        for p in range(4):
            assert np.array_equal(bits(got[i, p]), bits(ref[p])), (i, p)
        for k, v in s1.items():
            tot[k] = tot.get(k, 0) + v
    assert st == tot
    assert (got[1, 1, 100:140] == 77.0).all()
    with pytest.raises(srz.SrzError):
        ctx.draw_batch(fr, planes, primitive=7)
    ctx.close()


def test_draw_batch_in_pieces_with_registered_planes_and_without_the_depth_readback(orc):
    """srz_draw_batch renders a batch in pieces of ~128 MB of planes and reads piece k back while piece k + 1 renders: 40 frames of
    1024 x 512 (8 MB each: three pieces of 16 / 16 / 8) in page-locked planes (srz_host_register) — every frame equals the oracle bit
    for bit, accumulate-mode frames in the middle of a piece included; frames flagged SRZ_NO_Z_READBACK leave the caller's depth
    plane untouched; srz_draw with registered planes and the flag likewise"""
    import srz
    ctx = srz.Context(0)
    ctx.texture_upload(0, scenes.spot_texture())
    uniq = [scenes.config2(i, size=512).tris[0] for i in range(6)]
    W, H = 1024, 512

    def wide(i, flags):  # config 2's 512 x 512 triangles on a 1024 x 512 frame (a frame object of its own: with_flags() mutates)
        return abi.Frame(W, H, scenes.EYE, scenes.LIGHTS, [(abi.SHADER_TEXTURE, 0, uniq[i % 6])], flags)
    n = 40
    acc, noz = {5, 17, 33}, {2, 17, 39}
    fr = [wide(i, (0 if i in acc else abi.FUSED_CLEAR) | (abi.NO_Z_READBACK if i in noz else 0)) for i in range(n)]
    planes = np.zeros((n, 4, H, W), np.float32)
    planes[:, 0] = np.inf
    for i in acc:
        planes[i, 0, 200:260] = 2.0 + i
        planes[i, 1:4, 200:260] = 11.0 + i
    init = planes.copy()
    ctx.host_register(planes)
    try:
        got, _ = ctx.draw_batch(fr, planes)
        for i, f in enumerate(fr):
            rc, ref, _ = orc.draw(wide(i, f.c.flags & ~abi.NO_Z_READBACK), tuple(init[i, p].copy() for p in range(4)) if i in acc else None)
            assert rc == 0
            for p in range(4):
                want = init[i, 0] if (p == 0 and i in noz) else ref[p]
                assert np.array_equal(bits(got[i, p]), bits(want)), (i, p)
        # srz_draw, registered planes, no depth read-back
        one = tuple(planes[0, p] for p in range(4))
        one[0][:] = -5.0
        ctx.draw(fr[2], one)
        rc, ref, _ = orc.draw(wide(2, abi.FUSED_CLEAR))
        assert (one[0] == -5.0).all() and all(np.array_equal(bits(one[p]), bits(ref[p])) for p in (1, 2, 3))
    finally:
        ctx.host_unregister(planes)
    ctx.close()


def test_undocumented_flag_bits_and_debug_environment_are_ignored(orc, frames, monkeypatch):
    """the kernel-ablation switches of the development builds are gone: neither SRZ_DEBUG_FLAGS nor stray high flag bits
    change a render"""
    import srz
    monkeypatch.setenv("SRZ_DEBUG_FLAGS", "1")
    ctx = srz.Context(0)
    ctx.texture_upload(0, scenes.spot_texture())
    _, a = render(ctx, frames)
    _, b = render(ctx, frames, flags=abi.FUSED_CLEAR | 0xff00)
    assert torch.equal(a.view(torch.int32), b.view(torch.int32))
    rc, ref, _ = orc.draw(frames[2])
    assert np.array_equal(bits(a[2].cpu().numpy()), bits(np.stack(ref)))
    ctx.close()


def test_srz_draw_reuses_its_frameset_between_calls(orc):
    """srz_draw keeps its device state while the structure of the calls is unchanged and re-uploads only the data; a change
    of structure (another size, another batch layout) rebuilds it.  Every call must equal the oracle."""
    import srz
    ctx = srz.Context(0)
    ctx.texture_upload(0, scenes.spot_texture())
    seq = [scenes.config2(1, size=256), scenes.config2(9, size=256), scenes.config2(9, size=256, shader=abi.SHADER_PHONG),
           scenes.config2(4, size=320), scenes.config2(5, size=320), scenes.config3(2, 320, 200), scenes.config2(6, size=256)]
    for i, f in enumerate(seq):
        rc, ref, rst = orc.draw(f)
        gpu, gst = ctx.draw(f, want_stats=(i % 2 == 0))
        for p in range(4):
            assert np.array_equal(bits(gpu[p]), bits(ref[p])), (i, p)
        assert gst is None or gst == rst
    # accumulate mode through the cached path: the second draw starts from the first one's planes
    f0, f1 = scenes.config2(0, size=256), scenes.config2(18, size=256, flags=0)
    rc, ref, _ = orc.draw(f0)
    rc, ref, _ = orc.draw(f1, ref)
    gpu, _ = ctx.draw(f0)
    gpu, _ = ctx.draw(f1, gpu)
    for p in range(4):
        assert np.array_equal(bits(gpu[p]), bits(ref[p])), p
    ctx.close()


@pytest.mark.gpu
def test_lane_renderer_equals_one_frameset_and_timing_samples():
    """a batch as 2 / 3 lanes (framesets on streams of their own) gives the bytes of the batch as one frameset, batch after
    batch; the context's per-render samples and their span are consistent"""
    import srz
    ctx = srz.Context(0)
    frames = [scenes.config2(i, size=256) for i in range(20)]
    ctx.texture_upload(scenes.TEX_SPOT, scenes.spot_texture())
    fs = ctx.frameset(frames)
    ref = torch.empty(fs.out_shape, dtype=torch.float32, device="cuda")
    fs.render(ref.data_ptr(), fs.out_bytes, abi.FUSED_CLEAR)
    torch.cuda.synchronize()
    for lanes in (2, 3):
        lr = parallel.LaneRenderer(ctx, frames, lanes)
        assert lr.cuts[0] == 0 and lr.cuts[-1] == 20 and all(a < b for a, b in zip(lr.cuts, lr.cuts[1:]))
        assert lr.out_shape == fs.out_shape and lr.out_bytes == fs.out_bytes
        out = torch.full(lr.out_shape, float("nan"), dtype=torch.float32, device="cuda")
        ctx.set_kernel_timing(1)
        ctx.kernel_time_ms(reset=True)
        for _ in range(3):  # back to back: the lanes overlap across batches
            lr.render(out.data_ptr(), abi.FUSED_CLEAR)
        lr.synchronize()
        samples, span = ctx.kernel_time_samples()
        kt = ctx.kernel_time_ms(reset=True)
        ctx.set_kernel_timing(0)
        assert torch.equal(out.view(torch.int32), ref.view(torch.int32))
        assert len(samples) == 3 * lanes == kt["launches"] and all(x > 0 for x in samples)
        assert max(samples) <= span * 1.001
        assert kt["bin_ms"] == 0 and abs(kt["total_ms"] - sum(samples) / len(samples)) < 1e-3
        st = lr.stats()  # (the texture term of the algorithmic bytes needs the counters)
        assert st == fs.stats()
        assert lr.algorithmic_bytes() == fs.algorithmic_bytes()
        lr.close()
    fs.close()


@pytest.mark.parametrize("n_frames", [3, 8, 19])
def test_mixed_fast_and_generic_frames_share_one_set(orc, n_frames):
    """k_shade's two builds take their tiles from different work lists of the same render: frames the FAST build may shade
    (2 lights, p = 150, NORMAL / TEXTURE / PHONG) interleaved with frames only the generic one can (3 lights, another
    exponent, an empty frame), at frame counts below, at and above the 8 lists' period — every frame bit-identical to the
    oracle's"""
    import srz
    ctx = srz.Context(0)
    ctx.texture_upload(scenes.TEX_SPOT, scenes.spot_texture())
    three = np.concatenate([scenes.LIGHTS, np.array([[[-0.7, 0.2, 0.8], [30, 40, 50]]], np.float32)])
    frames = []
    for i in range(n_frames):
        base = scenes.config2(i, size=256, shader=(abi.SHADER_TEXTURE, abi.SHADER_PHONG, abi.SHADER_NORMAL)[i % 3])
        tris = base.tris[0]
        if i % 4 == 1:      # generic: three lights
            frames.append(abi.Frame(256, 256, scenes.EYE, three, [(abi.SHADER_TEXTURE, scenes.TEX_SPOT, tris)], abi.FUSED_CLEAR))
        elif i % 4 == 3:    # generic: another exponent
            frames.append(abi.Frame(256, 256, scenes.EYE, scenes.LIGHTS, [(abi.SHADER_PHONG, -1, tris)], abi.FUSED_CLEAR, p=8.0))
        elif i == 4:        # nothing to draw: every tile is the fused clear's
            frames.append(abi.Frame(256, 256, scenes.EYE, scenes.LIGHTS, [(abi.SHADER_NORMAL, -1, tris[:0])], abi.FUSED_CLEAR))
        else:
            frames.append(base)
    fs, out = render(ctx, frames)
    got = out.cpu().numpy()
    for i, f in enumerate(frames):
        rc, ref, st = orc.draw(f)
        for p in range(4):
            assert np.array_equal(bits(got[i, p]), bits(ref[p])), (n_frames, i, p)
    fs.close()
    ctx.close()


def test_two_lanes_at_batch_size_take_turns_and_match_the_oracle(orc):
    """two lanes of 8 frames of 1024^2 each (8192 tiles per lane: the side-stream clear and the raster turns between streams
    are both active), four batches back to back into rotating outputs: every frame of every batch is the oracle's"""
    import srz
    ctx = srz.Context(0)
    ctx.texture_upload(scenes.TEX_SPOT, scenes.spot_texture())
    frames = [scenes.config2(i) for i in range(16)]
    lr = parallel.LaneRenderer(ctx, frames, 2)
    assert lr.cuts == [0, 8, 16]
    outs = [torch.full(lr.out_shape, float("nan"), dtype=torch.float32, device="cuda") for _ in range(2)]
    for k in range(4):
        lr.render(outs[k % 2].data_ptr(), abi.FUSED_CLEAR)
    lr.synchronize()
    ref = {}
    for i in (0, 5, 8, 15):
        rc, planes, st = orc.draw(frames[i])
        ref[i] = planes
    for o in outs:
        got = o.cpu().numpy()
        for i, planes in ref.items():
            for p in range(4):
                assert np.array_equal(bits(got[i, p]), bits(planes[p])), (i, p)
    assert torch.equal(outs[0].view(torch.int32), outs[1].view(torch.int32))
    lr.close()
    ctx.close()


def test_clear_grid_is_measured_and_every_grid_gives_the_same_bits(orc):
    """a batch-sized set (8 frames of 1024^2: the clear runs on the side stream) measures the grid of its clear within its first 24
    renders, on the device (srz_device.h, ClearCtl; srz_api.hip, srz_frameset::ClearTune): every one of them — each candidate grid, frames of one plane per work item — leaves
    the oracle's bits, untouched tiles included (the buffer is poisoned before every render), and after the measurement the set
    reports a grid out of the candidates"""
    import srz
    ctx = srz.Context(0)
    ctx.texture_upload(scenes.TEX_SPOT, scenes.spot_texture())
    frames = [scenes.config2(i) for i in range(8)]
    fs = ctx.frameset(frames)
    assert fs.debug_counters()["clear_tuned"] == 0
    s = torch.cuda.current_stream().cuda_stream
    out = torch.empty(fs.out_shape, dtype=torch.float32, device="cuda")
    first = None
    for k in range(26):
        out.fill_(float("nan"))
        fs.render(out.data_ptr(), fs.out_bytes, abi.FUSED_CLEAR, s)
        torch.cuda.synchronize()
        if first is None:
            first = out.clone()
            got = first.cpu().numpy()
            for i in (0, 7):
                ref = orc.draw(frames[i])[1]
                for p in range(4):
                    assert np.array_equal(bits(got[i, p]), bits(ref[p])), (i, p)
        else:
            assert torch.equal(out.view(torch.int32), first.view(torch.int32)), k
    dc = fs.debug_counters()
    assert dc["clear_tuned"] == 1 and dc["clear_wgs"] in (96, 128, 256), dc
    # a second set of the same shape in the same ctx starts with the grid the first one measured
    fs2 = ctx.frameset(frames)
    out.fill_(float("nan"))
    fs2.render(out.data_ptr(), fs2.out_bytes, abi.FUSED_CLEAR, s)
    torch.cuda.synchronize()
    dc2 = fs2.debug_counters()
    assert dc2["clear_tuned"] == 1 and dc2["clear_wgs"] == dc["clear_wgs"], (dc, dc2)
    assert torch.equal(out.view(torch.int32), first.view(torch.int32))
    fs2.close()
    # ... and again 4096 renders later (18 renders, no synchronisation in between): still the same bits, and a grid once more
    for k in range(4096 + 20):
        fs.render(out.data_ptr(), fs.out_bytes, abi.FUSED_CLEAR, s)
    torch.cuda.synchronize()
    out.fill_(float("nan"))
    fs.render(out.data_ptr(), fs.out_bytes, abi.FUSED_CLEAR, s)
    torch.cuda.synchronize()
    dc = fs.debug_counters()
    assert dc["clear_tuned"] == 1 and dc["clear_wgs"] in (96, 128, 256), dc
    assert torch.equal(out.view(torch.int32), first.view(torch.int32))
    fs.close()
    ctx.close()


def test_lane_renderer_default_flags_are_the_fused_clear(orc):
    """LaneRenderer.render(ptr) with no flags must mean what FrameSet.render(ptr, bytes) means: the buffer counts as just
    cleared.  (Round 2's default was SRZ_UNIFIED = accumulate with 8-wide semantics: a torch.empty buffer then fed garbage
    depths into the draw.)"""
    import srz
    ctx = srz.Context(0)
    ctx.texture_upload(scenes.TEX_SPOT, scenes.spot_texture())
    frames = [scenes.config2(i, size=256) for i in range(16)]
    lr = parallel.LaneRenderer(ctx, frames, 2)
    out = torch.full(lr.out_shape, float("nan"), dtype=torch.float32, device="cuda")  # NaN depths: poison for an accumulating draw
    lr.render(out.data_ptr())
    lr.synchronize()
    got = out.cpu().numpy()
    for i in (0, 7, 8, 15):
        ref = orc.draw(frames[i])[1]
        for p in range(4):
            assert np.array_equal(bits(got[i, p]), bits(ref[p])), (i, p)
    lr.close()
    ctx.close()


def test_large_set_rendered_as_sub_batches_matches_the_oracle(orc, monkeypatch):
    """a large set is rendered as sub-batches of whole frames one after the other (shared counters, pool and work lists; the
    size is 256 frames' worth of 1024^2 in tiles — forced to 104 frames here, small frames being cheap to check): 328 frames,
    twice in a row, every frame the oracle's"""
    import srz
    monkeypatch.setenv("SRZ_SUB_BATCH", "104")
    ctx = srz.Context(0)
    ctx.texture_upload(scenes.TEX_SPOT, scenes.spot_texture())
    uniq = [scenes.config2(i, size=96, shader=(abi.SHADER_TEXTURE, abi.SHADER_PHONG)[i % 2]) for i in range(36)]
    frames = [uniq[(7 * i) % 36] for i in range(328)]
    fs = ctx.frameset(frames)
    out = torch.full(fs.out_shape, float("nan"), dtype=torch.float32, device="cuda")
    for _ in range(2):
        fs.render(out.data_ptr(), fs.out_bytes, abi.FUSED_CLEAR, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    got = out.cpu().numpy()
    ref = {}
    for i in range(328):
        k = (7 * i) % 36
        if k not in ref:
            ref[k] = orc.draw(uniq[k])[1]
        for p in range(4):
            assert np.array_equal(bits(got[i, p]), bits(ref[k][p])), (i, p)
    fs.close()
    ctx.close()

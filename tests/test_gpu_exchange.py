"""-m gpu: the multi-GPU exchange behind the C ABI (srz_comm_*, srz_frameset_allgather, srz_frameset_deinterleave).

A single GPU can play every rank of an N-GPU job: each rank's band shard is rendered with its own sharded context, the
all-gather's result (rank-major concatenation) is assembled by hand, and the HIP de-interleave pass must return the
oracle's full frames bit for bit.  The RCCL communicator itself is exercised with world = 1 here and with world = 2 when
two GPUs are visible."""
import os
import socket

import numpy as np
import pytest
import torch

import scenes
from srz import abi, parallel

pytestmark = pytest.mark.gpu


def bits(a):
    return np.ascontiguousarray(a).view(np.uint32)


@pytest.mark.parametrize("world,size", [(2, 256), (3, 320), (8, 512)])
def test_deinterleave_kernel_restores_row_major_frames(orc, world, size):
    import srz
    frames = [scenes.config2(i, size=size) for i in (2, 11, 25)]
    refs = [np.stack(orc.draw(f)[1]) for f in frames]
    shards, shards8, fs_keep = [], [], []
    for r in range(world):
        ctx = srz.Context(0, r, world)
        ctx.texture_upload(0, scenes.spot_texture())
        fs = ctx.frameset(frames)
        out = torch.zeros(fs.out_shape, dtype=torch.float32, device="cuda")
        s = torch.cuda.current_stream().cuda_stream
        fs.render(out.data_ptr(), fs.out_bytes, abi.FUSED_CLEAR, s)
        b8 = torch.zeros((len(frames), 1, fs.local_rows, size * 3), dtype=torch.uint8, device="cuda")
        fs.resolve8(out.data_ptr(), b8.data_ptr(), b8.numel(), s)
        torch.cuda.synchronize()
        shards.append(out), shards8.append(b8), fs_keep.append((ctx, fs))
    ctx, fs = fs_keep[0]
    bpr = fs.local_rows // 32
    gathered = torch.stack(shards).contiguous()          # = what ncclAllGather leaves: [rank][frame][plane][rows][W]
    full = torch.empty((len(frames), 4, bpr * world * 32, size), dtype=torch.float32, device="cuda")
    fs.deinterleave(gathered.data_ptr(), full.data_ptr(), abi.EXCHANGE_PLANES, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    got = full.cpu().numpy()[:, :, :size]
    for i in range(len(frames)):
        assert np.array_equal(bits(got[i]), bits(refs[i])), i
    assert torch.equal(full, parallel.deinterleave(gathered, world))                       # = the torch formulation
    g8 = torch.stack(shards8).contiguous()
    full8 = torch.empty((len(frames), 1, bpr * world * 32, size * 3), dtype=torch.uint8, device="cuda")
    fs.deinterleave(g8.data_ptr(), full8.data_ptr(), abi.EXCHANGE_BGR8, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    for i in range(len(frames)):
        ref8 = orc.resolve8(tuple(refs[i]))
        assert np.array_equal(full8[i, 0, :size].cpu().numpy().reshape(size, size, 3), ref8), i
    for c, f in fs_keep:
        f.close(), c.close()


def test_communicator_of_one_rank(orc):
    """srz_comm_unique_id / srz_comm_create / srz_frameset_allgather through librccl with world = 1"""
    import srz
    ctx = srz.Context(0)
    ctx.texture_upload(0, scenes.spot_texture())
    comm = srz.Comm(ctx, srz.Comm.unique_id(), 0, 1)
    frames = [scenes.config2(i, size=256) for i in (1, 2)]
    fs = ctx.frameset(frames)
    assert fs.exchange_bytes(abi.EXCHANGE_PLANES) == fs.out_bytes and fs.exchange_bytes(abi.EXCHANGE_BGR8) == 2 * 256 * 256 * 3
    out = torch.zeros(fs.out_shape, dtype=torch.float32, device="cuda")
    gathered, full = torch.empty_like(out), torch.full_like(out, -7.0)
    s = torch.cuda.Stream()
    fs.render(out.data_ptr(), fs.out_bytes, abi.FUSED_CLEAR, s.cuda_stream)
    fs.allgather(comm, out.data_ptr(), gathered.data_ptr(), full.data_ptr(), abi.EXCHANGE_PLANES, s.cuda_stream)
    s.synchronize()
    for i, f in enumerate(frames):
        ref = np.stack(orc.draw(f)[1])
        assert np.array_equal(bits(full[i].cpu().numpy()), bits(ref))
    with pytest.raises(srz.SrzError):   # a frameset of another shard
        ctx2 = srz.Context(0, 1, 2)
        fs2 = ctx2.frameset(frames)
        fs2.allgather(comm, out.data_ptr(), gathered.data_ptr(), full.data_ptr())
    comm.close()
    ctx.close()


@pytest.mark.parametrize("exchange,layout", [("planes", "shards"), ("bgr8", "shards"), ("planes", "rows"), ("bgr8", "rows")])
def test_bench_multi_gpu_step_loop_at_world_one(orc, exchange, layout):
    """bench.py's N > 1 branch (render stream + exchange stream, double-buffered ExchangePipeline, srz_frameset_allgather
    through a real RCCL communicator, the exchange-alone pass) driven at world 1, so that the 8-GPU scaling run is not that
    code's first execution; the last step's full frames must be the oracle's."""
    import bench
    import srz
    ctx = srz.Context(0)
    comm = srz.Comm(ctx, srz.Comm.unique_id(), 0, 1)
    case = bench.Case(ctx, torch, "spot_texture_1024", 6, "raster", 1, n_out=2)
    for slot, tex in enumerate(scenes_textures(case)):
        orc.texture_set(slot, tex)

    def fence():
        torch.cuda.synchronize()

    dt, kt, per_step, multi = bench.time_multi_gpu(case, comm, None, steps=3, warmup=2, fence=fence, exchange=exchange, layout=layout)
    assert dt > 0 and kt["launches"] == 3 and multi["behind_c_abi"] and multi["overlapped"] and multi["second_pass"] == (layout == "rows")
    assert multi["bytes_sent_per_rank_per_step"] == case.fs.exchange_bytes(abi.EXCHANGE_PLANES if exchange == "planes" else abi.EXCHANGE_BGR8)
    full = multi.pop("last_full").cpu().numpy()
    if layout == "shards":
        full = full[0]  # (one rank: its shard is the frame)
    import json
    json.dumps(multi)  # (what is left goes into bench.py's JSON line)
    for i in (0, 5):
        ref = orc.draw(case.frames[i])[1]
        if exchange == "planes":
            assert np.array_equal(bits(full[i]), bits(np.stack(ref)))
        else:
            assert np.array_equal(full[i, 0].reshape(1024, 1024, 3), orc.resolve8(tuple(ref)))
    case.close()
    comm.close()
    ctx.close()


def scenes_textures(case):
    from srz import scenes as pscenes
    wl = pscenes.WORKLOADS[case.name]()
    wl.frame(0)
    return wl.texture_arrays


@pytest.mark.parametrize("cfg,frames_idx", [(4, (3,)), (5, (3,))])
def test_baseline_configs_4_and_5_as_specified_eight_way_sharded(orc, cfg, frames_idx):
    """BASELINE configs 4 (spot x16, 2048^2, TEXTURE) and 5 (8 depth-stacked spots, 4096^2, NORMAL + PHONG) AS SPECIFIED:
    every frame's 32-row bands dealt to 8 ranks.  One GPU plays every rank in turn; the stacked shards are what the all-gather
    leaves, and srz_frameset_deinterleave of them must be the oracle's full frames bit for bit — float planes and display()'s
    8-bit image."""
    import srz
    world = 8
    build = {4: scenes.config4, 5: scenes.config5}[cfg]
    frames = [build(i) for i in frames_idx]
    size = frames[0].width
    refs = [np.stack(orc.draw(f)[1]) for f in frames]
    lay = parallel.shard_layout(size, 0, world)
    gathered = torch.empty((world, len(frames), 4, lay["local_rows"], size), dtype=torch.float32, device="cuda")
    g8 = torch.empty((world, len(frames), 1, lay["local_rows"], size * 3), dtype=torch.uint8, device="cuda")
    keep = None
    for r in range(world):
        ctx = srz.Context(0, r, world)
        ctx.texture_upload(0, scenes.spot_texture())
        fs = ctx.frameset(frames)
        assert fs.local_rows == lay["local_rows"]
        s = torch.cuda.current_stream().cuda_stream
        fs.render(gathered[r].data_ptr(), fs.out_bytes, abi.FUSED_CLEAR, s)
        fs.resolve8(gathered[r].data_ptr(), g8[r].data_ptr(), g8[r].numel(), s)
        torch.cuda.synchronize()
        if r == 0:
            keep = (ctx, fs)
        else:
            fs.close(), ctx.close()
    ctx, fs = keep
    rows = lay["bands_per_rank"] * world * 32
    full = torch.empty((len(frames), 4, rows, size), dtype=torch.float32, device="cuda")
    full8 = torch.empty((len(frames), 1, rows, size * 3), dtype=torch.uint8, device="cuda")
    s = torch.cuda.current_stream().cuda_stream
    fs.deinterleave(gathered.data_ptr(), full.data_ptr(), abi.EXCHANGE_PLANES, s)
    fs.deinterleave(g8.data_ptr(), full8.data_ptr(), abi.EXCHANGE_BGR8, s)
    torch.cuda.synchronize()
    for i in range(len(frames)):
        got = full[i, :, :size].cpu().numpy()
        assert np.array_equal(bits(got), bits(refs[i])), (cfg, i)
        assert np.array_equal(full8[i, 0, :size].cpu().numpy().reshape(size, size, 3), orc.resolve8(tuple(refs[i]))), (cfg, i)
    # the exchange WITHOUT the second pass leaves exactly `gathered` (every rank rendered into its own slot above): rows are
    # found by srz_frameset_gathered_row_offset, and a frame comes to the host de-interleaved by the copies themselves
    gflat, g8flat = gathered.view(torch.uint8).reshape(-1), g8.reshape(-1)
    for (p, y) in ((0, 0), (1, 31), (2, 32), (3, size - 1), (0, 32 * 8 + 5), (2, 32 * 13 + 17)):
        off = fs.gathered_row_offset(0, p, y)
        row = gflat[off: off + size * 4].view(torch.float32).cpu().numpy()
        assert np.array_equal(bits(row), bits(refs[0][p][y])), (cfg, p, y)
        assert torch.equal(parallel.gathered_row(gathered, 0, p, y, world), gathered.view(torch.uint8).reshape(-1)[off: off + size * 4].view(torch.float32))
    off8 = fs.gathered_row_offset(0, 0, 77, abi.EXCHANGE_BGR8)
    assert np.array_equal(g8flat[off8: off8 + size * 3].cpu().numpy().reshape(size, 3), orc.resolve8(tuple(refs[0]))[77])
    for bad in ((0, 4, 0, abi.EXCHANGE_PLANES), (0, 0, size, abi.EXCHANGE_PLANES), (len(frames), 0, 0, abi.EXCHANGE_PLANES), (0, 0, 0, 7)):
        with pytest.raises(IndexError):   # out of range / unknown exchange kind: the C sentinel (size_t)-1 never reaches a pointer
            fs.gathered_row_offset(bad[0], bad[1], bad[2], bad[3])
    host = fs.read_gathered_frame(gathered.data_ptr(), 0)
    assert np.array_equal(bits(host), bits(refs[0])), cfg
    host8 = fs.read_gathered_frame(g8.data_ptr(), 0, abi.EXCHANGE_BGR8)
    assert np.array_equal(host8, orc.resolve8(tuple(refs[0]))), cfg
    fs.close(), ctx.close()


def test_read_gathered_frame_with_a_short_last_band(orc):
    """a frame height that is not a multiple of 32 (1080 = 33 bands + 24 rows), three ranks: the strided device→host copies of
    srz_frameset_read_gathered_frame must place every band, the short one included"""
    import srz
    world, w, h = 3, 320, 200
    frames = [scenes.config3(4, w, h)]
    ref = np.stack(orc.draw(frames[0])[1])
    lay = parallel.shard_layout(h, 0, world)
    gathered = torch.zeros((world, 1, 4, lay["local_rows"], w), dtype=torch.float32, device="cuda")
    for r in range(world):
        ctx = srz.Context(0, r, world)
        ctx.texture_upload(0, scenes.spot_texture())
        fs = ctx.frameset(frames)
        fs.render(gathered[r].data_ptr(), fs.out_bytes, abi.FUSED_CLEAR, torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        if r + 1 < world:
            fs.close(), ctx.close()
    host = fs.read_gathered_frame(gathered.data_ptr(), 0)
    assert np.array_equal(bits(host), bits(ref))
    fs.close(), ctx.close()


def test_render_on_the_null_stream_is_ordered(orc):
    """stream 0 (torch's default stream) must mean the NULL stream, not the context's private non-blocking stream: a torch
    op queued right after the render has to see its result."""
    import srz
    ctx = srz.Context(0)
    ctx.texture_upload(0, scenes.spot_texture())
    f = scenes.config2(4, size=512)
    fs = ctx.frameset([f] * 8)
    out = torch.zeros(fs.out_shape, dtype=torch.float32, device="cuda")
    assert torch.cuda.current_stream().cuda_stream == 0
    for _ in range(3):
        out.fill_(-3.0)
        fs.render(out.data_ptr(), fs.out_bytes, abi.FUSED_CLEAR, 0)
        copy = out.clone()          # torch's null stream: ordered after the render only if the render ran there
    torch.cuda.synchronize()
    ref = np.stack(orc.draw(f)[1])
    assert np.array_equal(bits(copy[7].cpu().numpy()), bits(ref))
    ctx.close()


def _two_rank_worker(rank, world, port, q):
    import conftest  # noqa: F401
    import torch.distributed as dist
    import srz
    from oracle import oracle
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    torch.cuda.set_device(rank)
    dist.init_process_group("gloo", rank=rank, world_size=world)   # (rendezvous only: the exchange is srz_comm's RCCL)
    try:
      try:
          oracle.texture_set(0, scenes.spot_texture())
          ctx = srz.Context(rank, rank, world)
          ctx.texture_upload(0, scenes.spot_texture())
          ids = [srz.Comm.unique_id() if rank == 0 else None]
          dist.broadcast_object_list(ids, src=0)
          comm = srz.Comm(ctx, ids[0], rank, world)
          size, nf, steps = 256, 3, 5
          sets = [ctx.frameset([scenes.config2(3 * k + i, size=size) for i in range(nf)]) for k in range(steps)]
          fs0 = sets[0]
          bpr = fs0.local_rows // 32
          shard = [torch.zeros(fs0.out_shape, dtype=torch.float32, device="cuda") for _ in range(2)]
          gathered = [torch.empty((world,) + tuple(fs0.out_shape), dtype=torch.float32, device="cuda") for _ in range(2)]
          full = [torch.empty((nf, 4, bpr * world * 32, size), dtype=torch.float32, device="cuda") for _ in range(2)]
          rq, xq = parallel.TorchQueue(), parallel.TorchQueue()
          state = {"k": 0}
          pipe = parallel.ExchangePipeline(
              lambda b: sets[state["k"]].render(shard[b].data_ptr(), fs0.out_bytes, abi.FUSED_CLEAR, rq.handle),
              lambda b: fs0.allgather(comm, shard[b].data_ptr(), gathered[b].data_ptr(), full[b].data_ptr(), abi.EXCHANGE_PLANES, xq.handle),
              rq, xq)
          ok, snaps = True, []
          for k in range(steps):   # a different set of frames every step: a torn or stale buffer cannot go unnoticed
              state["k"] = k
              b = pipe.step()
              with torch.cuda.stream(xq.stream):
                  snaps.append(full[b].clone())
          pipe.drain()
          torch.cuda.synchronize()
          # the exchange without the second pass: render into the own slot of the gathered buffer, one in-place all-gather
          g = torch.zeros((world,) + tuple(fs0.out_shape), dtype=torch.float32, device="cuda")
          sets[1].render(g[rank].data_ptr(), fs0.out_bytes, abi.FUSED_CLEAR, rq.handle)
          rq.drain()
          fs0.allgather_inplace(comm, g.data_ptr(), abi.EXCHANGE_PLANES, xq.handle)
          xq.drain()
          for i in range(nf):
              ref = np.stack(oracle.draw(scenes.config2(3 + i, size=size))[1])
              ok = ok and np.array_equal(fs0.read_gathered_frame(g.data_ptr(), i).view(np.uint32), ref.view(np.uint32))
          for k in range(steps):
              got = snaps[k].cpu().numpy()[:, :, :size]
              for i in range(nf):
                  ref = np.stack(oracle.draw(scenes.config2(3 * k + i, size=size))[1])
                  ok = ok and np.array_equal(got[i].view(np.uint32), ref.view(np.uint32))
          q.put((rank, ok))
          comm.close()
      except BaseException as e:  # noqa: BLE001  (the parent must not wait 300 s for a worker that died)
        q.put((rank, f"worker failed: {e!r}"))
        raise
    finally:
        dist.destroy_process_group()


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs (RCCL refuses two ranks on one device)")
def test_two_ranks_overlapped_exchange_equals_oracle():
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mctx = mp.get_context("spawn")
    q = mctx.Queue()
    procs = [mctx.Process(target=_two_rank_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in range(2)]
    for p in procs:
        p.join(timeout=60)
    assert all(ok is True for _, ok in res), res


@pytest.mark.parametrize("w,h,world", [(101, 67, 1), (101, 67, 3), (322, 70, 2), (37, 33, 1)])
def test_device_resolve_and_bgr8_exchange_for_any_width(orc, w, h, world):
    """srz_frameset_resolve8 and the 8-bit exchange's de-interleave used to refuse widths that are not a multiple of 4 (rows of W x 3
    bytes that are no whole dwords): every size now resolves on the device — unsharded (the plane size decides between the 4-pixel
    and the one-pixel kernel) and sharded (one GPU plays every rank; the byte-granular de-interleave restores the rows)"""
    import srz
    frames = [scenes.config3(i, w, h) for i in (2, 5)]
    refs = [orc.draw(f)[1] for f in frames]
    lay = parallel.shard_layout(h, 0, world)
    rows = lay["local_rows"]
    g8 = torch.zeros((world, len(frames), 1, rows, w * 3), dtype=torch.uint8, device="cuda")
    planes = torch.zeros((world, len(frames), 4, rows, w), dtype=torch.float32, device="cuda")
    keep = None
    for r in range(world):
        ctx = srz.Context(0, r, world)
        fs = ctx.frameset(frames)
        assert fs.local_rows == rows
        s = torch.cuda.current_stream().cuda_stream
        fs.render(planes[r].data_ptr(), fs.out_bytes, abi.FUSED_CLEAR, s)
        fs.resolve8(planes[r].data_ptr(), g8[r].data_ptr(), g8[r].numel(), s)
        torch.cuda.synchronize()
        if r == 0:
            keep = (ctx, fs)
        else:
            fs.close(), ctx.close()
    ctx, fs = keep
    if world == 1:
        for i, ref in enumerate(refs):
            assert np.array_equal(g8[0, i, 0].cpu().numpy().reshape(h, w, 3), orc.resolve8(tuple(ref))), (w, h, i)
    else:
        full8 = torch.zeros((len(frames), 1, lay["bands_per_rank"] * world * 32, w * 3), dtype=torch.uint8, device="cuda")
        fs.deinterleave(g8.data_ptr(), full8.data_ptr(), abi.EXCHANGE_BGR8, torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        for i, ref in enumerate(refs):
            assert np.array_equal(full8[i, 0, :h].cpu().numpy().reshape(h, w, 3), orc.resolve8(tuple(ref))), (w, h, world, i)
            host8 = fs.read_gathered_frame(g8.data_ptr(), i, abi.EXCHANGE_BGR8)
            assert np.array_equal(host8, orc.resolve8(tuple(ref)))
    fs.close(), ctx.close()

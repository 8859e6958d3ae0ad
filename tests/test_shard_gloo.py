"""not-gpu: the N>1 path (band sharding → all-gather → de-interleave) with world_size 2 and 3 over gloo on the CPU.
Each rank fills its shard with the ORACLE restricted to its own 32-row bands (the checker standing in for the kernels);
after the all-gather every rank must hold exactly the full oracle frame."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, height, width, q):
    import conftest  # noqa: F401
    import scenes
    from oracle import oracle
    from srz import parallel
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        oracle.texture_set(scenes.TEX_SPOT, scenes.spot_texture())
        frames = [scenes.config2(i, size=width) if height == width else scenes.config3(i, width, height) for i in (1, 4)]
        lay = parallel.shard_layout(height, rank, world)
        shard = torch.zeros((len(frames), 4, lay["local_rows"], width), dtype=torch.float32)
        full_ref = []
        for fi, f in enumerate(frames):
            planes = oracle.new_planes(width, height)
            for (lb, band, r0, r1) in parallel.band_rows(height, rank, world):
                assert oracle.draw_rows(f, planes, r0, r1) == 0
                for p in range(4):
                    shard[fi, p, lb * 32: lb * 32 + (r1 - r0)] = torch.from_numpy(planes[p][r0:r1])
            rc, ref, _ = oracle.draw(f)
            full_ref.append(np.stack(ref))
        gathered = torch.empty((world,) + tuple(shard.shape), dtype=torch.float32)
        full = parallel.all_gather_frames(shard, world, gathered)
        got = full.numpy()[:, :, :height]
        ok = all(np.array_equal(got[i].view(np.uint32), full_ref[i].view(np.uint32)) for i in range(len(frames)))

        # the 8-bit exchange of bench.py --exchange bgr8: display()'s resolve per shard, same all-gather on uint8 rows
        def to_bgr8(planes_f32):  # [frames, 4, rows, W] float → [frames, 1, rows, W*3] uint8 (round half to even, saturate)
            c = torch.nan_to_num(planes_f32[:, 1:4], nan=0.0).round().clamp(0, 255).to(torch.uint8)
            return c.permute(0, 2, 3, 1).reshape(c.shape[0], 1, c.shape[2], -1).contiguous()
        shard8 = to_bgr8(shard)
        gathered8 = torch.empty((world,) + tuple(shard8.shape), dtype=torch.uint8)
        full8 = parallel.all_gather_frames(shard8, world, gathered8)
        ref8 = to_bgr8(torch.from_numpy(np.stack(full_ref)))
        ok = ok and torch.equal(full8[:, :, :height], ref8) and bool((ref8 > 0).any())
        # the exchange without the second pass: the shard sits in its own slot of the gathered buffer, ONE in-place all-gather,
        # rows located by the rank-major rule of srz_frameset_gathered_row_offset
        g2 = torch.zeros((world,) + tuple(shard.shape), dtype=torch.float32)
        g2[rank] = shard
        parallel.all_gather_inplace(g2, rank)
        for fi in range(len(frames)):
            for p in range(4):
                for y in (0, 31, 32, 63, height - 1, height // 2):
                    ok = ok and np.array_equal(parallel.gathered_row(g2, fi, p, y, world).numpy().view(np.uint32), full_ref[fi][p][y].view(np.uint32))
        q.put((rank, ok, tuple(full.shape)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,height,width", [(2, 256, 256), (3, 200, 320)])
def test_band_sharding_allgather_reassembles_the_frame(world, height, width):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, height, width, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert sorted(r[0] for r in res) == list(range(world))
    assert all(r[1] for r in res), res


def _pipeline_worker(rank, world, port, q):
    """the overlapped render → exchange pipeline of bench.py (srz.parallel.ExchangePipeline) with two worker threads standing
    in for the two HIP streams: different frames every step, every step's gathered frames checked before the buffer is reused"""
    import conftest  # noqa: F401
    import scenes
    from oracle import oracle
    from srz import parallel
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        oracle.texture_set(scenes.TEX_SPOT, scenes.spot_texture())
        size, nf, steps = 128, 2, 5
        lay = parallel.shard_layout(size, rank, world)
        shard = [torch.zeros((nf, 4, lay["local_rows"], size), dtype=torch.float32) for _ in range(2)]
        gathered = [torch.empty((world,) + tuple(shard[0].shape), dtype=torch.float32) for _ in range(2)]
        full = [torch.empty((nf, 4, lay["bands_per_rank"] * world * 32, size), dtype=torch.float32) for _ in range(2)]
        bad = []

        def render(b):
            k = state_of[b]
            for fi in range(nf):
                f = scenes.config2(2 * k + fi, size=size)
                planes = oracle.new_planes(size, size)
                for (lb, band, r0, r1) in parallel.band_rows(size, rank, world):
                    assert oracle.draw_rows(f, planes, r0, r1) == 0
                    for p in range(4):
                        shard[b][fi, p, lb * 32: lb * 32 + (r1 - r0)] = torch.from_numpy(planes[p][r0:r1])

        def exchange(b):
            parallel.all_gather_frames(shard[b], world, gathered[b], full[b])

        def check(k, b):
            for fi in range(nf):
                ref = np.stack(oracle.draw(scenes.config2(2 * k + fi, size=size))[1])
                if not np.array_equal(full[b][fi].numpy()[:, :size].view(np.uint32), ref.view(np.uint32)):
                    bad.append((k, fi))

        state_of = {}
        rq, xq = parallel.ThreadQueue(), parallel.ThreadQueue()
        pipe = parallel.ExchangePipeline(render, exchange, rq, xq)
        for k in range(steps):
            rq.submit(lambda k=k: state_of.__setitem__(k % 2, k))  # (on the render queue: in order with render(b))
            b = pipe.step()
            xq.submit(lambda k=k, b=b: check(k, b))
        pipe.drain()
        rq.close(), xq.close()
        q.put((rank, not bad, bad))
    finally:
        dist.destroy_process_group()


def test_overlapped_exchange_pipeline_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_pipeline_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in range(2)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(r[1] for r in res), res


def test_shard_layout_matches_the_c_side_rules():
    from srz import parallel
    for h in (32, 33, 200, 1024, 1080, 4096):
        for world in (1, 2, 3, 8):
            owned = []
            for r in range(world):
                lay = parallel.shard_layout(h, r, world)
                rows = parallel.band_rows(h, r, world)
                assert len(rows) == lay["n_local_bands"] <= lay["bands_per_rank"]
                assert lay["local_rows"] == (h if world == 1 else lay["bands_per_rank"] * 32)
                owned += [b for (_, b, _, _) in rows]
            assert sorted(owned) == list(range((h + 31) // 32))   # every band owned exactly once


def test_band_map_is_a_partition_for_every_world_and_height():
    """the band → rank map (srz/parallel.py = band_of / rank_of_band in csrc/srz_device.h: every round of `world` consecutive bands hands one
    band to every rank, rotated by five ranks per round — one where five is a multiple of the world): every band has exactly one owner,
    a rank's local band index is the round's index, counts differ by at most one, and the torch de-interleave agrees with the row rule"""
    from srz import parallel
    for world in (1, 2, 3, 4, 5, 7, 8, 10):
        for height in (1, 31, 32, 33, 70, 513, 1000, 1024, 2048, 4096):
            n_bands = (height + 31) // 32
            owner = {}
            counts = []
            for r in range(world):
                lay = parallel.shard_layout(height, r, world)
                rows = parallel.band_rows(height, r, world)
                assert len(rows) == lay["n_local_bands"] <= lay["bands_per_rank"] == (n_bands + world - 1) // world
                counts.append(len(rows))
                for lb, b, r0, r1 in rows:
                    assert b not in owner and parallel.rank_of_band(b, world) == r and b // world == lb and (r0, r1) == (b * 32, min(height, b * 32 + 32))
                    owner[b] = r
            assert sorted(owner) == list(range(n_bands)) and max(counts) - min(counts) <= 1, (world, height)
    assert parallel.band_rot(8) == 5 and parallel.band_rot(5) == 1 and parallel.band_rot(1) == 1
    # a rank meets a different slice of a 16-band object in consecutive rounds (what the rotation is for)
    assert {parallel.band_of(lb, 3, 8) % 16 for lb in range(8)} == {(3 - 5 * lb) % 8 + 8 * (lb % 2) for lb in range(8)}
    for world, bpr in ((3, 4), (8, 5), (5, 3)):
        g = torch.arange(world * 2 * 4 * bpr * 32 * 3, dtype=torch.float32).view(world, 2, 4, bpr * 32, 3)
        full = parallel.deinterleave(g, world)
        for row in range(0, bpr * world * 32, 5):
            assert torch.equal(full[1, 2, row], parallel.gathered_row(g, 1, 2, row, world))

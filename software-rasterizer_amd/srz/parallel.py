"""Multi-GPU band sharding: one process per GPU, 32-row bands dealt round-robin with a rotation per round (band_of), RCCL all-gather to reassemble.

Mirrors shard_layout() in csrc/srz_api.hip.  Works on any torch.distributed backend (nccl = RCCL on the GPUs, gloo in
the CPU tests)."""
from . import abi

BAND = 32


BAND_ROT = 5  # = SRZ_BAND_ROT in csrc/srz_device.h


def band_rot(world):
    return BAND_ROT if BAND_ROT % world else 1


def band_of(lb, rank, world):
    """band of local band `lb` of `rank`: every group of `world` consecutive bands hands one band to every rank, rotated by
    band_rot(world) steps per group (= band_of in csrc/srz_device.h, where the choice is measured)"""
    return lb * world + (rank - band_rot(world) * lb) % world


def rank_of_band(b, world):
    return (b + band_rot(world) * (b // world)) % world


def shard_layout(height, rank, world):
    """→ dict(n_bands, n_local_bands, bands_per_rank, local_rows) for the ctx of (rank, world)."""
    n_bands = (height + BAND - 1) // BAND
    n_local = n_bands // world
    if band_of(n_local, rank, world) < n_bands:
        n_local += 1
    per_rank = (n_bands + world - 1) // world
    local_rows = height if world == 1 else per_rank * BAND
    return {"n_bands": n_bands, "n_local_bands": n_local, "bands_per_rank": per_rank, "local_rows": local_rows}


def band_rows(height, rank, world):
    """[(local_band, band, row0, row1)] of the bands owned by `rank`."""
    lay = shard_layout(height, rank, world)
    out = []
    for lb in range(lay["n_local_bands"]):
        b = band_of(lb, rank, world)
        out.append((lb, b, b * BAND, min(height, (b + 1) * BAND)))
    return out


def deinterleave(gathered, world, out=None):
    """gathered: [world, frames, 4, bands_per_rank*32, W] (all-gather of every rank's shard) → row-major
    [frames, 4, bands_per_rank*world*32, W]; rows >= height are all-gather padding."""
    w_, n_frames, planes, local_rows, width = gathered.shape
    assert w_ == world and local_rows % BAND == 0
    bpr = local_rows // BAND
    import torch
    src = gathered.view(world, n_frames, planes, bpr, BAND, width)
    if out is None:
        out = torch.empty((n_frames, planes, bpr * world * BAND, width), dtype=gathered.dtype, device=gathered.device)
    dst = out.view(n_frames, planes, bpr, world, BAND, width)  # [.., group g, position j in the group, ..]: band g * world + j
    for j in range(world):  # position j of group g belongs to rank (j + rot * g) % world, whose local band g it is
        ranks = (band_rot(world) * torch.arange(bpr, device=gathered.device) + j) % world
        dst[:, :, :, j] = src[ranks, :, :, torch.arange(bpr, device=gathered.device)].permute(1, 2, 0, 3, 4)
    return out


def gathered_row(gathered, frame, plane, row, world):
    """the row `row` of a frame inside a rank-major gathered buffer [world, frames, planes, bands_per_rank*32, W] (the layout
    srz_frameset_allgather_inplace leaves; = srz_frameset_gathered_row_offset)"""
    band = row // BAND
    return gathered[rank_of_band(band, world), frame, plane, (band // world) * BAND + row % BAND]


def all_gather_inplace(gathered, rank, group=None):
    """the exchange without a second pass: gathered[rank] already holds this rank's shard; one in-place all-gather"""
    import torch.distributed as dist
    dist.all_gather_into_tensor(gathered.view(-1), gathered[rank].reshape(-1), group=group)
    return gathered


def all_gather_frames(shard, world, gathered, full=None, group=None):
    """shard: this rank's [frames,4,local_rows,W] tensor → every rank gets every full frame."""
    import torch.distributed as dist
    dist.all_gather_into_tensor(gathered.view(-1), shard.reshape(-1), group=group)
    return deinterleave(gathered, world, full)


class TorchQueue:
    """An in-order device queue for ExchangePipeline: a torch (HIP) stream."""

    def __init__(self, stream=None):
        import torch
        self.torch, self.stream = torch, stream if stream is not None else torch.cuda.Stream()

    @property
    def handle(self):
        return self.stream.cuda_stream

    def submit(self, fn):
        with self.torch.cuda.stream(self.stream):
            fn()

    def record(self):
        ev = self.torch.cuda.Event()
        ev.record(self.stream)
        return ev

    def wait(self, ev):
        self.stream.wait_event(ev)

    def drain(self):
        self.stream.synchronize()


class ThreadQueue:
    """The same contract on the CPU (a worker thread): lets the tests run the overlapped pipeline with real asynchrony."""

    def __init__(self):
        import queue
        import threading
        self._q, self._threading, self.error = queue.Queue(), threading, None
        self._t = threading.Thread(target=self._run, daemon=True)
        self._t.start()

    def _run(self):
        while True:
            fn = self._q.get()
            if fn is None:
                return
            try:
                fn()
            except BaseException as e:  # noqa: BLE001  (reported by drain)
                self.error = e

    def submit(self, fn):
        self._q.put(fn)

    def record(self):
        ev = self._threading.Event()
        self._q.put(ev.set)
        return ev

    def wait(self, ev):
        self._q.put(ev.wait)

    def drain(self):
        self.record().wait()
        if self.error is not None:
            raise self.error

    def close(self):
        self._q.put(None)


class ExchangePipeline:
    """render → exchange, double-buffered on two in-order queues so that the exchange of step k (all-gather over xGMI +
    de-interleave) runs while step k+1 renders:

        step k uses buffer b = k % n_buffers
        render queue  : wait(exchange of step k - n_buffers done: shard[b] / full[b] are free) → render(b)
        exchange queue: wait(render of step k done)                                           → exchange(b)

    render(b) / exchange(b) submit asynchronous work to the CURRENT queue (TorchQueue: the torch stream is current; pass
    queue.handle to the C ABI).  After drain() the result of step k is in full[k % n_buffers]."""

    def __init__(self, render, exchange, render_q, exchange_q, n_buffers=2):
        self.render, self.exchange, self.rq, self.xq, self.n = render, exchange, render_q, exchange_q, n_buffers
        self.ev_x = [None] * n_buffers
        self.k = 0

    def step(self):
        b = self.k % self.n
        if self.ev_x[b] is not None:
            self.rq.wait(self.ev_x[b])
        self.rq.submit(lambda: self.render(b))
        ev_r = self.rq.record()
        self.xq.wait(ev_r)
        self.xq.submit(lambda: self.exchange(b))
        self.ev_x[b] = self.xq.record()
        self.k += 1
        return b

    def drain(self):
        self.rq.drain()
        self.xq.drain()


class LaneRenderer:
    """A batch rendered as `lanes` runs of whole frames, each a frameset with a stream of its own.

    Within a lane the renders are in stream order; the lanes are NOT synchronised with each other, so when batches are
    submitted back to back the lanes drift out of phase and the kernels of one fill the launch gaps and draining tails of
    the other's (k_raster is LDS-bound, k_shade VALU-bound, k_clear HBM-bound).  MI355X, 256 frames of 1024^2: 1.15 ms per
    batch on one stream, 1.07 ms on two lanes.  Splitting ONE render into slices gains nothing (measured: both slices are in
    the same phase at the same time); the gain is the overlap of consecutive batches, which only the caller can allow —
    hence a host-side helper over two srz_frameset handles, not a mode of srz_frameset_render.

    out: [n_frames, 4, local_rows, W] float32 on the device; lane k writes frames [f0_k, f0_k+1)."""

    def __init__(self, ctx, frames, lanes=2, streams=None):
        import torch
        frames = list(frames)
        n = len(frames)
        lanes = max(1, min(int(lanes), n))
        # whole frames, lane boundaries on multiples of 8 where the batch allows (the kernels deal frames to the 8 XCDs by 8)
        cuts = [0]
        for k in range(1, lanes):
            c = n * k // lanes
            c8 = (c + 7) // 8 * 8
            cuts.append(c8 if cuts[-1] < c8 < n else max(c, cuts[-1] + 1))
        cuts.append(n)
        self.ctx, self.cuts, self.n_frames = ctx, cuts, n
        self.sets = [ctx.frameset(frames[cuts[k]:cuts[k + 1]]) for k in range(lanes)]
        self.streams = list(streams) if streams is not None else [torch.cuda.Stream() for _ in range(lanes)]
        assert len(self.streams) == lanes
        fs0 = self.sets[0]
        self.width, self.height, self.local_rows = fs0.width, fs0.height, fs0.local_rows
        self.frame_bytes = 16 * self.local_rows * self.width
        self.out_bytes = self.frame_bytes * n

    @property
    def out_shape(self):
        return (self.n_frames, 4, self.local_rows, self.width)

    def render(self, d_out_ptr, flags=abi.FUSED_CLEAR):
        """one batch: every lane renders its frames into its part of out, asynchronously, on its own stream.  flags as
        FrameSet.render: the default treats the buffer as just cleared (write-only); 0 = accumulate onto what out holds"""
        for k, fs in enumerate(self.sets):
            fs.render(d_out_ptr + self.cuts[k] * self.frame_bytes, fs.out_bytes, flags, self.streams[k].cuda_stream)

    def wait(self, stream):
        """make `stream` (a torch stream) wait for everything submitted to the lanes so far"""
        for q in self.streams:
            stream.wait_stream(q)

    def synchronize(self):
        for q in self.streams:
            q.synchronize()

    def stats(self):
        tot = {}
        for fs in self.sets:
            for k, v in fs.stats().items():
                tot[k] = tot.get(k, 0) + v
        return tot

    def algorithmic_bytes(self):
        return sum(fs.algorithmic_bytes() for fs in self.sets)

    def close(self):
        for fs in self.sets:
            fs.close()
        self.sets = []

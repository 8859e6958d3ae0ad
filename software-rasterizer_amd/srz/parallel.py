"""Multi-GPU band sharding: one process per GPU, 32-row bands dealt round-robin, RCCL all-gather to reassemble.

Mirrors shard_layout() in csrc/srz_api.hip.  Works on any torch.distributed backend (nccl = RCCL on the GPUs, gloo in
the CPU tests)."""
BAND = 32


def shard_layout(height, rank, world):
    """→ dict(n_bands, n_local_bands, bands_per_rank, local_rows) for the ctx of (rank, world)."""
    n_bands = (height + BAND - 1) // BAND
    n_local = (n_bands - rank + world - 1) // world if rank < n_bands else 0
    per_rank = (n_bands + world - 1) // world
    local_rows = height if world == 1 else per_rank * BAND
    return {"n_bands": n_bands, "n_local_bands": n_local, "bands_per_rank": per_rank, "local_rows": local_rows}


def band_rows(height, rank, world):
    """[(local_band, band, row0, row1)] of the bands owned by `rank`."""
    lay = shard_layout(height, rank, world)
    out = []
    for lb in range(lay["n_local_bands"]):
        b = lb * world + rank
        out.append((lb, b, b * BAND, min(height, (b + 1) * BAND)))
    return out


def deinterleave(gathered, world, out=None):
    """gathered: [world, frames, 4, bands_per_rank*32, W] (all-gather of every rank's shard) → row-major
    [frames, 4, bands_per_rank*world*32, W]; rows >= height are all-gather padding."""
    w_, n_frames, planes, local_rows, width = gathered.shape
    assert w_ == world and local_rows % BAND == 0
    bpr = local_rows // BAND
    src = gathered.view(world, n_frames, planes, bpr, BAND, width).permute(1, 2, 3, 0, 4, 5)
    if out is None:
        return src.reshape(n_frames, planes, bpr * world * BAND, width)
    out.view(n_frames, planes, bpr, world, BAND, width).copy_(src)
    return out


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

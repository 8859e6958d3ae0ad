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

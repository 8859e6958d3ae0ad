"""Dev tool: per-tile timeline of k_raster (stats variant) → concurrency analysis."""
import os
import sys
os.environ.setdefault("SRZ_DEBUG_FLAGS", "1")
import conftest  # noqa
import numpy as np
import torch
torch.cuda.init()
import scenes, srz
F = int(sys.argv[1]) if len(sys.argv) > 1 else 64
uniq = [scenes.config2(i) for i in range(min(F, 36))]
frames = [uniq[i % len(uniq)] for i in range(F)]
ctx = srz.Context(0); ctx.texture_upload(0, scenes.spot_texture())
fs = ctx.frameset(frames)
fs.stats()
n = F * 32 * 32
from srz import abi
dbg = int(sys.argv[2], 0) if len(sys.argv) > 2 else 0
out = torch.empty(fs.out_shape, dtype=torch.float32, device='cuda')
for _ in range(2):
    fs.render(out.data_ptr(), fs.out_bytes, abi.FUSED_CLEAR | dbg)
torch.cuda.synchronize()
ctx.debug_timeline(n, True)
fs.render(out.data_ptr(), fs.out_bytes, abi.FUSED_CLEAR | dbg)
torch.cuda.synchronize()
ctx.sync()
tl = ctx.debug_timeline(n, False).astype(np.int64)
t0 = tl[:, 0][tl[:, 0] > 0].min()
st, en, hw, blocks = (tl[:, 0] - t0) / 100.0, (tl[:, 1] - t0) / 100.0, tl[:, 2], tl[:, 3]   # us
dur = en - st
heavy = blocks > 0
print(f"tiles {n} heavy {heavy.sum()} kernel span {en.max():.1f} us; heavy dur mean {dur[heavy].mean():.1f} max {dur[heavy].max():.1f} us; light dur mean {dur[~heavy].mean():.2f} us")
print(f"us per hit triangle (heavy): {(dur[heavy].sum()/blocks[heavy].sum()):.3f}; start time of last heavy: {st[heavy].max():.1f}, of last tile {st.max():.1f}")
ts = np.linspace(0, en.max(), 21)
for t in ts:
    print(f"  t={t:7.1f}us resident={int(((st<=t)&(en>t)).sum()):5d} heavy_resident={int(((st<=t)&(en>t)&heavy).sum()):5d} dispatched={int((st<=t).sum())}")
cu = (hw >> 8) & 0xf; sh = (hw >> 12) & 1; se = (hw >> 13) & 0x7; simd = (hw >> 4) & 0x3; 
print("simd distribution (heavy):", np.bincount(simd[heavy], minlength=4))
print("distinct (se,sh,cu):", len(set(zip(se[heavy], sh[heavy], cu[heavy]))))

"""Dev tool: where does the order-independent rasteriser differ from the oracle?"""
import sys
import numpy as np
import conftest  # noqa
import scenes
import srz
from srz import abi
from oracle import oracle
oracle.texture_set(0, scenes.spot_texture())
ctx = srz.Context(0)
ctx.texture_upload(0, scenes.spot_texture())
size = int(sys.argv[1]) if len(sys.argv) > 1 else 512
f = scenes.config2(5, size=size, shader=abi.SHADER_NORMAL)
rc, ref, _ = oracle.draw(f)
gpu, _ = ctx.draw(f)
gz, rz = gpu[0], ref[0]
bad = gz.view(np.uint32) != rz.view(np.uint32)
print("differ", bad.sum(), "ref finite", np.isfinite(rz).sum(), "gpu finite", np.isfinite(gz).sum())
print("gpu inf where ref finite", (np.isinf(gz) & np.isfinite(rz)).sum(), "gpu finite where ref inf", (np.isfinite(gz) & np.isinf(rz)).sum())
both = np.isfinite(gz) & np.isfinite(rz) & bad
print("both finite but differ", both.sum())
ys, xs = np.nonzero(bad)
print("rows", ys.min(), ys.max(), "cols", xs.min(), xs.max())
for y, x in list(zip(ys, xs))[:12]:
    print(y, x, gz[y, x], rz[y, x], [gpu[p][y, x] for p in (1, 2, 3)], [ref[p][y, x] for p in (1, 2, 3)])
tiles = {}
for y, x in zip(ys // 32, xs // 32):
    tiles[(y, x)] = tiles.get((y, x), 0) + 1
print("bad tiles", len(tiles), list(tiles.items())[:10])
ry, rx = np.nonzero(np.isfinite(rz))
print("ref covered tiles", len(set(zip(ry // 32, rx // 32))))

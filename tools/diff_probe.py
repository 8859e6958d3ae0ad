"""dev tool: GPU vs oracle on one frame; prints per-plane mismatch counts and the first few differing pixels"""
import sys
import conftest  # noqa
import numpy as np
import scenes, srz
from oracle import oracle
from srz import abi
cfg = sys.argv[1] if len(sys.argv) > 1 else "2"
size = int(sys.argv[2]) if len(sys.argv) > 2 else 256
oracle.texture_set(scenes.TEX_SPOT, scenes.spot_texture())
ctx = srz.Context(0)
ctx.texture_upload(scenes.TEX_SPOT, scenes.spot_texture())
f = {"2": lambda: scenes.config2(13, size=size), "2p": lambda: scenes.config2(13, size=size, shader=abi.SHADER_PHONG),
     "2n": lambda: scenes.config2(13, size=size, shader=abi.SHADER_NORMAL)}[cfg]()
rc, ref, rst = oracle.draw(f)
gpu, gst = ctx.draw(f, want_stats=True)
print("stats equal", gst == rst)
for p in range(4):
    bad = np.ascontiguousarray(gpu[p]).view(np.uint32) != np.ascontiguousarray(ref[p]).view(np.uint32)
    print("plane", p, "mismatches", int(bad.sum()), "of covered", int(np.isfinite(ref[0]).sum()))
    for (y, x) in np.argwhere(bad)[:5]:
        print("   ", y, x, "gpu", gpu[p][y, x], "ref", ref[p][y, x])

"""dev tool: one frame of the fuzz test on the GPU vs the oracle, mismatches listed per batch"""
import sys
import conftest  # noqa
import numpy as np
import torch
import scenes, srz
from oracle import oracle
from srz import abi
import test_gpu_frameset as T
seed, fi = int(sys.argv[1]), int(sys.argv[2])
rng = np.random.default_rng(1000 + seed)
w, h = [(64, 64), (200, 120), (97, 131), (256, 96), (33, 290), (128, 128), (320, 200), (70, 70)][seed % 8]
flags = abi.FUSED_CLEAR | (abi.UNIFIED if seed % 3 == 2 else 0)
frames = [T._random_frame(rng, w, h, int(rng.integers(1, 400)), flags) for _ in range(int(rng.integers(2, 12)))]
oracle.texture_set(0, scenes.spot_texture())
ctx = srz.Context(0)
ctx.texture_upload(0, scenes.spot_texture())
f = frames[fi]
print("lights", f.c.n_lights, "p", f.c.p, "batches", [(f._batches[b].shader, f._batches[b].n_tris) for b in range(f.c.n_batches)])
for b in range(f.c.n_batches):
    fb = abi.Frame(w, h, tuple(f.c.eye), f.lights, [(f._batches[b].shader, f._batches[b].tex_id, f.tris[b])], flags, p=f.c.p)
    fsb = ctx.frameset([fb])
    out = torch.zeros(fsb.out_shape, dtype=torch.float32, device="cuda")
    fsb.render(out.data_ptr(), fsb.out_bytes, 0, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    g = out.cpu().numpy()[0]
    r = oracle.draw(fb)[1]
    bad = g[1].view(np.uint32) != np.ascontiguousarray(r[1]).view(np.uint32)
    print("batch", b, "shader", f._batches[b].shader, "mismatches", int(bad.sum()), "visible", int(np.isfinite(r[0]).sum()))
    for (y, x) in np.argwhere(bad)[:6]:
        print("    ", y, x, "gpu", g[1][y, x], g[2][y, x], g[3][y, x], "ref", r[1][y, x], r[2][y, x], r[3][y, x])

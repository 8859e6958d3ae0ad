"""dev tool: one 256-frame set on one stream vs S slices of 256/S frames on S streams (do complementary kernels of different
slices fill each other's gaps and tails?)"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "software-rasterizer_amd"))
import torch
import srz
from srz import abi, scenes

name = sys.argv[1] if len(sys.argv) > 1 else "spot_texture_1024"
F = int(sys.argv[2]) if len(sys.argv) > 2 else 256
n = int(sys.argv[3]) if len(sys.argv) > 3 else 60
wl = scenes.WORKLOADS[name]()
ctx = srz.Context(0)
uniq = [wl.frame(i) for i in range(36)]
wl.upload_textures(ctx)
frames = [uniq[i % 36] for i in range(F)]
for S, n_streams in ((1, 1), (2, 1), (2, 2), (4, 2), (4, 4), (8, 2), (1, 1)):
    per = F // S
    sets = [ctx.frameset(frames[k * per:(k + 1) * per]) for k in range(S)]
    outs = [torch.empty(fs.out_shape, dtype=torch.float32, device="cuda") for fs in sets]
    streams = [torch.cuda.Stream() for _ in range(n_streams)]
    def step():
        for k, fs in enumerate(sets):
            fs.render(outs[k].data_ptr(), fs.out_bytes, abi.FUSED_CLEAR, streams[k % n_streams].cuda_stream)
    for _ in range(15):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    print(f"{name} F={F} slices={S} streams={n_streams}: {dt*1e3:.4f} ms/step  {F/dt:.0f} frames/s", flush=True)
    for fs in sets:
        fs.close()
    del outs
    torch.cuda.empty_cache()

"""dev tool: the PCIe-inclusive rate of the boundary's host-buffer entry points (srz_draw: planes in / out as host memory) against the
device-resident frameset path — config 2 (spot TEXTURE 1024^2)"""
import time
import torch  # (first: its bundled HIP runtime must initialise the device before libsrz.so's does)
torch.cuda.init()
import conftest  # noqa: F401
import numpy as np
import scenes
import srz
from srz import abi

ctx = srz.Context(0)
ctx.texture_upload(scenes.TEX_SPOT, scenes.spot_texture())
frames = [scenes.config2(i) for i in range(36)]
planes = tuple(np.zeros((1024, 1024), np.float32) for _ in range(4))
for f in frames[:4]:
    ctx.draw(f, planes)
n = 200
t0 = time.perf_counter()
for i in range(n):
    ctx.draw(frames[i % 36], planes)            # SRZ_FUSED_CLEAR: 0.56 MB of triangles up, 16.8 MB of planes down per frame
dt = time.perf_counter() - t0
print(f"srz_draw, fused clear (download only): {n / dt:.0f} frames/s, {dt / n * 1e3:.3f} ms per frame, {16.78e6 * n / dt / 1e9:.1f} GB/s of planes over PCIe")
acc = [f.with_flags(0) for f in [scenes.config2(i) for i in range(8)]]
t0 = time.perf_counter()
for i in range(n // 2):
    ctx.draw(acc[i % 8], planes)                # in/out: 16.8 MB up + 16.8 MB down
dt2 = time.perf_counter() - t0
print(f"srz_draw, in/out planes (upload + download): {n // 2 / dt2:.0f} frames/s, {dt2 / (n // 2) * 1e3:.3f} ms per frame")
batch = frames[:32]
out = np.zeros((32, 4, 1024, 1024), np.float32)
out[:, 0] = np.inf
ctx.draw_batch(batch, out)
t0 = time.perf_counter()
for _ in range(5):
    ctx.draw_batch(batch, out)
dt3 = (time.perf_counter() - t0) / 5
print(f"srz_draw_batch, 32 frames per call (host planes): {32 / dt3:.0f} frames/s, {dt3 * 1e3:.1f} ms per call")

# ---- the same with the caller's planes page-locked once (srz_host_register): DMA at the link's rate -----------------------------------
for a in planes:
    ctx.host_register(a)
for f in frames[:4]:
    ctx.draw(f, planes)
t0 = time.perf_counter()
for i in range(n):
    ctx.draw(frames[i % 36], planes)
dt = time.perf_counter() - t0
print(f"srz_draw, REGISTERED planes, fused clear (download only): {n / dt:.0f} frames/s, {dt / n * 1e3:.3f} ms per frame, {16.78e6 * n / dt / 1e9:.1f} GB/s down")
noz = [scenes.config2(i, flags=abi.FUSED_CLEAR | abi.NO_Z_READBACK) for i in range(8)]
t0 = time.perf_counter()
for i in range(n):
    ctx.draw(noz[i % 8], planes)
dt = time.perf_counter() - t0
print(f"srz_draw, REGISTERED planes, fused clear, SRZ_NO_Z_READBACK: {n / dt:.0f} frames/s, {dt / n * 1e3:.3f} ms per frame, {12.58e6 * n / dt / 1e9:.1f} GB/s down")
t0 = time.perf_counter()
for i in range(n // 2):
    ctx.draw(acc[i % 8], planes)
dt2 = time.perf_counter() - t0
print(f"srz_draw, REGISTERED planes, in/out (upload + download): {n // 2 / dt2:.0f} frames/s, {dt2 / (n // 2) * 1e3:.3f} ms per frame, "
      f"{16.78e6 * (n // 2) / dt2 / 1e9:.1f} GB/s each way if serial ({2 * 16.78e6 * (n // 2) / dt2 / 1e9:.1f} GB/s moved)")
for a in planes:
    ctx.host_unregister(a)
ctx.host_register(out)
ctx.draw_batch(batch, out)
t0 = time.perf_counter()
for _ in range(5):
    ctx.draw_batch(batch, out)
dt3 = (time.perf_counter() - t0) / 5
print(f"srz_draw_batch, 32 frames per call, REGISTERED planes: {32 / dt3:.0f} frames/s, {dt3 * 1e3:.1f} ms per call, {32 * 16.78e6 / dt3 / 1e9:.1f} GB/s down")
accb = [scenes.config2(i, flags=0) for i in range(32)]
out[:, 0] = np.inf
out[:, 1:] = 0
t0 = time.perf_counter()
for _ in range(3):
    ctx.draw_batch(accb, out)
dt4 = (time.perf_counter() - t0) / 3
print(f"srz_draw_batch, 32 in/out frames per call, REGISTERED planes: {32 / dt4:.0f} frames/s, {dt4 * 1e3:.1f} ms per call, {32 * 16.78e6 / dt4 / 1e9:.1f} GB/s each way")
ctx.host_unregister(out)

# ---- what the link itself gives: plain pinned copies of one frame's planes, no render --------------------------------------------------
import torch
h = torch.empty(4 * 1024 * 1024, dtype=torch.float32).pin_memory()
d = torch.empty(4 * 1024 * 1024, dtype=torch.float32, device="cuda")
for name, src, dst in (("D2H", d, h), ("H2D", h, d)):
    for _ in range(5):
        dst.copy_(src, non_blocking=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(100):
        dst.copy_(src, non_blocking=True)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 100
    print(f"raw link, pinned {name} of 16.8 MB: {dt * 1e3:.3f} ms, {16.78e6 / dt / 1e9:.1f} GB/s")

"""dev tool: what a pure streaming fill reaches on this GPU (the render writes 16 B per pixel: its practical floor)"""
import time, torch
n = 256 * 4 * 1024 * 1024  # floats: 256 frames of 1024^2 x 4 planes = 4.29 GB
x = torch.empty(n, dtype=torch.float32, device="cuda")
y = torch.empty(n, dtype=torch.float32, device="cuda")
for name, fn in (("zero_", lambda: x.zero_()), ("fill_(inf)", lambda: x.fill_(float("inf"))), ("copy_", lambda: y.copy_(x))):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        fn()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    b = n * 4 * (2 if name == "copy_" else 1)
    print(f"{name}: {ms:.3f} ms  {b/ms/1e9:.2f} TB/s", flush=True)

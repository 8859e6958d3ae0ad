"""dev tool: per-render launch-set times in sequence (is there a clock ramp / periodic slow step?)"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "software-rasterizer_amd"))
import torch
import srz
from srz import abi, scenes

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100
wl = scenes.spot_texture_1024()
ctx = srz.Context(0)
uniq = [wl.frame(i) for i in range(36)]
wl.upload_textures(ctx)
fs = ctx.frameset([uniq[i % 36] for i in range(256)])
out = torch.empty(fs.out_shape, dtype=torch.float32, device="cuda")
st = torch.cuda.Stream()
ctx.set_kernel_timing(1)
ctx.kernel_time_ms(True)
t0 = time.perf_counter()
for _ in range(n):
    fs.render(out.data_ptr(), fs.out_bytes, abi.FUSED_CLEAR, st.cuda_stream)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
s = ctx.kernel_time_samples()
print("wall/step %.4f ms" % (dt / n * 1e3))
for i in range(0, n, 10):
    print(i, " ".join("%.3f" % x for x in s[i:i + 10]))

"""dev tool: does the KIND of device allocation of the output planes matter?  (default vs uncached vs fine-grained)"""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "software-rasterizer_amd"))
import torch
import srz
from srz import abi, scenes

hip = C.CDLL("libamdhip64.so")
hip.hipExtMallocWithFlags.argtypes = [C.POINTER(C.c_void_p), C.c_size_t, C.c_uint]
hip.hipFree.argtypes = [C.c_void_p]
wl = scenes.spot_texture_1024()
ctx = srz.Context(0)
uniq = [wl.frame(i) for i in range(36)]
wl.upload_textures(ctx)
fs = ctx.frameset([uniq[i % 36] for i in range(256)])
st = torch.cuda.Stream()
for name, flag in (("default", 0x0), ("finegrained", 0x1), ("uncached", 0x3), ("default", 0x0)):
    p = C.c_void_p()
    rc = hip.hipExtMallocWithFlags(C.byref(p), fs.out_bytes, flag)
    if rc != 0:
        print(name, "alloc failed", rc); continue
    for _ in range(15):
        fs.render(p.value, fs.out_bytes, abi.FUSED_CLEAR, st.cuda_stream)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(50):
        fs.render(p.value, fs.out_bytes, abi.FUSED_CLEAR, st.cuda_stream)
    torch.cuda.synchronize()
    print(f"{name:12s} {(time.perf_counter()-t0)/50*1e3:.4f} ms per 256 frames", flush=True)
    hip.hipFree(p)

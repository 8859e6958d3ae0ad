"""Dev experiment: two half-batches on two streams (their kernels may overlap) vs one full batch."""
import sys, time
import conftest  # noqa
import torch
import srz
from srz import abi, scenes as ps
F = int(sys.argv[1]) if len(sys.argv) > 1 else 256
wl = ps.spot_texture_1024()
uniq = [wl.frame(i) for i in range(36)]
def mk(n, off):
    ctx = srz.Context(0)
    wl.upload_textures(ctx)
    fs = ctx.frameset([uniq[(off + i) % 36] for i in range(n)])
    out = torch.empty(fs.out_shape, dtype=torch.float32, device="cuda")
    return ctx, fs, out, torch.cuda.Stream()
full = mk(F, 0)
halves = [mk(F // 2, 0), mk(F // 2, F // 2)]
quarters = [mk(F // 4, i * F // 4) for i in range(4)]
def run(sets, iters=20, stagger=False):
    for _ in range(3):
        for (c, fs, out, s) in sets:
            fs.render(out.data_ptr(), fs.out_bytes, abi.FUSED_CLEAR, s.cuda_stream)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        for (c, fs, out, s) in sets:
            fs.render(out.data_ptr(), fs.out_bytes, abi.FUSED_CLEAR, s.cuda_stream)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters * 1e3
print("full batch   : %.3f ms per %d frames" % (run([full]), F))
print("2 halves ||  : %.3f ms" % run(halves))
print("4 quarters ||: %.3f ms" % run(quarters))
print("full batch   : %.3f ms" % run([full]))

def run_staggered(sets, delay_us, iters=30):
    torch.cuda.synchronize()
    s2 = sets[1][3]
    with torch.cuda.stream(s2):
        torch.cuda._sleep(int(delay_us * 2300))  # ~2.3 GHz cycles
    t0 = time.perf_counter()
    for _ in range(iters):
        for (c, fs, out, s) in sets:
            fs.render(out.data_ptr(), fs.out_bytes, abi.FUSED_CLEAR, s.cuda_stream)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0 - delay_us * 1e-6) / iters * 1e3
for d in (0, 100, 200, 300, 400, 500):
    print("2 halves, second stream delayed %3d us: %.3f ms" % (d, run_staggered(halves, d)))

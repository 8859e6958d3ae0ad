"""Dev tool (not a test, not the bench): time the frameset path on the GPU for a config.
usage: python tools/perf_probe.py [config=2] [frames=64] [iters=20]"""
import os
import sys
import time

import conftest  # noqa: F401  (sys.path)
import numpy as np
import torch

import scenes
import srz
from srz import abi

cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 2
F = int(sys.argv[2]) if len(sys.argv) > 2 else 64
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 20
dbg = int(sys.argv[4], 0) if len(sys.argv) > 4 else 0  # extra render flags, e.g. 4 = ORDERED_RASTER
BASE_FLAGS = int(os.environ.get('PROBE_FLAGS', str(abi.FUSED_CLEAR)), 0)
builder = {2: scenes.config2, 3: scenes.config3, 4: scenes.config4, 5: scenes.config5}[cfg]
t0 = time.time()
import os
kw = {}
if os.environ.get('PROBE_SHADER') is not None and cfg == 2:
    kw['shader'] = int(os.environ['PROBE_SHADER'])
if os.environ.get('PROBE_FRAME_FLAGS') is not None:
    kw['flags'] = int(os.environ['PROBE_FRAME_FLAGS'], 0)
uniq = [builder(i, **kw) for i in range(min(F, 36))]
frames = [uniq[i % len(uniq)] for i in range(F)]
print(f"built {len(uniq)} frames in {time.time()-t0:.1f}s; tris/frame={frames[0].n_tris}")
ctx = srz.Context(0)
ctx.texture_upload(scenes.TEX_SPOT, scenes.spot_texture())
if os.environ.get('PROBE_APPROX'):  # the tolerance mode of the shaders (SRZ_OPT_APPROX_SHADE)
    ctx.set_option(abi.OPT_APPROX_SHADE, 1)
fs = ctx.frameset(frames)
st = fs.stats()
print("stats", st)
out = torch.empty(fs.out_shape, dtype=torch.float32, device="cuda")
_sk = os.environ.get('PROBE_STREAM', 'null')  # null | new | high : which stream the renders go to
_ts = torch.cuda.current_stream() if _sk == 'null' else torch.cuda.Stream(priority=-1 if _sk == 'high' else 0)
stream = _ts.cuda_stream
ctx.set_kernel_timing(int(os.environ.get('PROBE_TIMING', '2')))
for _ in range(3):
    fs.render(out.data_ptr(), fs.out_bytes, BASE_FLAGS | dbg, stream)
torch.cuda.synchronize()
ctx.kernel_time_ms(True)
t0 = time.time()
for _ in range(iters):
    fs.render(out.data_ptr(), fs.out_bytes, BASE_FLAGS | dbg, stream)
torch.cuda.synchronize()
dt = (time.time() - t0) / iters
kt = ctx.kernel_time_ms(True)
ab = fs.algorithmic_bytes()
print(f"F={F} wall/render={dt*1e3:.3f} ms  " + " ".join(f"{k}={v:.3f}" if k != "launches" else f"{k}={v}" for k, v in kt.items()))
print(f"frames/s={F/dt:.0f}  Mfrag/s={st['fragments']/dt/1e6:.1f}  algorithmic bytes={ab/1e6:.1f} MB  "
      f"achieved={ab/(max(kt['total_ms'], 1e-9)*1e-3)/1e9:.1f} GB/s (pipeline, events)  {ab/dt/1e9:.1f} GB/s (wall)")

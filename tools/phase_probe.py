"""Dev tool (GPU): where a k_shade wave's time per tile goes.  Needs the probe build: tools/mkvariant.sh probe -DSRZ_PHASE_PROBE, then
SRZ_LIB_PATH=software-rasterizer_amd/build/probe.so python tools/phase_probe.py <config> <frames>
Phases (shader clocks per wave per tile): 0 tile start → owner ids + list indices arrived (a wait that also covers the previous tile's
stores), 1 → classification + first barrier, 2 → compaction + staged triangles landed + second barrier, 3 → dense passes + barrier,
4 → write-out issued."""
import sys
import conftest  # noqa: F401
import torch
import scenes
import srz
from srz import abi

cfg, F = int(sys.argv[1]), int(sys.argv[2])
builder = {2: scenes.config2, 3: scenes.config3, 4: scenes.config4, 5: scenes.config5}[cfg]
uniq = [builder(i) for i in range(min(F, 12))]
ctx = srz.Context(0)
ctx.texture_upload(scenes.TEX_SPOT, scenes.spot_texture())
fs = ctx.frameset([uniq[i % len(uniq)] for i in range(F)])
out = torch.empty(fs.out_shape, dtype=torch.float32, device="cuda")
for _ in range(3):
    fs.render(out.data_ptr(), fs.out_bytes, abi.FUSED_CLEAR, torch.cuda.current_stream().cuda_stream)
torch.cuda.synchronize()
ctx.debug_counters()
n = 5
for _ in range(n):
    fs.render(out.data_ptr(), fs.out_bytes, abi.FUSED_CLEAR, torch.cuda.current_stream().cuda_stream)
torch.cuda.synchronize()
c = ctx.debug_counters()
ph = c[-6:]
tiles = ph[5] or 1
tot = sum(ph[:5])
names = ["ids+indices wait", "classify+barrier1", "compact+DMA+barrier2", "dense passes", "write-out issue"]
print(f"config {cfg}, {F} frames: {tiles / n / 4:.0f} tiles per render, {tot / tiles:.0f} clocks per wave-tile; "
      f"k_shade's waves ran at {c[0] / max(c[1], 1) * 100:.0f} MHz (s_memtime / s_memrealtime x 100 MHz)")
for k in range(5):
    print(f"  {names[k]:22s} {ph[k] / tiles:8.0f} clocks  {100 * ph[k] / tot:5.1f} %")

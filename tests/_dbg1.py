import sys
sys.path.insert(0,'tests')
import conftest, scenes, srz, numpy as np
ctx = srz.Context(0)
ctx.texture_upload(0, scenes.spot_texture())
f = scenes.config1()
gpu, st = ctx.draw(f, want_stats=True)
print(st, ctx.debug_counters())
print(np.isfinite(gpu[0]).sum())

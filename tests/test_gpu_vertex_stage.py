"""-m gpu: the device vertex stage (k_vertex = Scene::loadTriangleStream on the GPU): meshes resident
in HBM + per-frame matrices must give the same framebuffer, bit for bit, as host-built post-MVP streams and the oracle."""
import numpy as np
import pytest
import torch

from srz import abi
from srz import scenes as pscenes

pytestmark = pytest.mark.gpu


def bits(a):
    return np.ascontiguousarray(a).view(np.uint32)


@pytest.mark.parametrize("make", [pscenes.spot_texture_1024, pscenes.spot_bunny_1080p])
def test_sceneset_equals_frameset_and_oracle(orc, make):
    import srz
    wl = make()
    ctx = srz.Context(0)
    idx = (0, 7, 19)
    frames = [wl.frame(i) for i in idx]
    wl.upload_textures(ctx)
    wl.upload_meshes(ctx)
    sframes = [wl.scene_frame(i) for i in idx]
    for slot, tex in enumerate(wl.texture_arrays):
        orc.texture_set(slot, tex)
    stream = torch.cuda.current_stream().cuda_stream
    outs = []
    for fr in (frames, sframes):
        fs = ctx.frameset(fr)
        out = torch.zeros(fs.out_shape, dtype=torch.float32, device="cuda")
        fs.render(out.data_ptr(), fs.out_bytes, abi.FUSED_CLEAR, stream)
        torch.cuda.synchronize()
        outs.append(out.cpu().numpy())
        st = fs.stats()
    assert np.array_equal(bits(outs[0]), bits(outs[1]))
    for k, f in enumerate(frames):
        rc, ref, _ = orc.draw(f)
        assert np.array_equal(bits(outs[1][k]), bits(np.stack(ref)))
    # single-frame entry point with host planes
    planes, st1 = ctx.draw(sframes[1], want_stats=True)
    rc, ref, rst = orc.draw(frames[1])
    assert st1 == rst
    for a, b in zip(planes, ref):
        assert np.array_equal(bits(a), bits(b))
    ctx.close()


def test_mesh_upload_argument_checks():
    import srz
    ctx = srz.Context(0)
    v = np.zeros((3, 8), np.float32)
    with pytest.raises(srz.SrzError):
        ctx.mesh_upload(0, v, np.array([[0, 1, 3]], np.uint32))      # index out of range
    with pytest.raises(srz.SrzError):
        ctx.mesh_upload(999, v, np.array([[0, 1, 2]], np.uint32))    # slot out of range
    wl = pscenes.spot_texture_1024()
    with pytest.raises(srz.SrzError):
        ctx.frameset([wl.scene_frame(0)])                            # mesh slot 0 never uploaded
    ctx.close()

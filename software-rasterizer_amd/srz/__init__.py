"""srz — Python binding (ctypes) of the C ABI in include/srz.h, the MI355X raster + fragment-shade stage.

The compute path is ONLY libsrz.so (hand-written gfx950 kernels).  There is no CPU fallback: if the
library is missing or no gfx950 device is present, every call raises.
"""
import ctypes as C
import os

import numpy as np

from . import abi
from .abi import Frame  # noqa: F401

_PKG = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB_PATH = os.environ.get("SRZ_LIB_PATH", os.path.join(_PKG, "libsrz.so"))  # (override: A/B of dev builds)
_lib = None

EXPORTS = ["srz_abi_version", "srz_create", "srz_destroy", "srz_last_error", "srz_set_shard", "srz_set_option", "srz_texture_upload",
           "srz_draw", "srz_draw_scene", "srz_mesh_upload", "srz_sceneset_create", "srz_frameset_create", "srz_frameset_destroy", "srz_frameset_local_rows",
           "srz_frameset_out_bytes", "srz_frameset_render", "srz_frameset_resolve8", "srz_frameset_stats", "srz_frameset_algorithmic_bytes",
           "srz_kernel_time_ms", "srz_kernel_time_samples", "srz_set_kernel_timing", "srz_sync", "srz_debug_counters", "srz_verify_fastmath", "srz_verify_fastdiv", "srz_verify_fastpow", "srz_verify_fastlen", "srz_host_register", "srz_host_unregister", "srz_frameset_debug_counters", "srz_draw_batch",
           "srz_comm_unique_id", "srz_comm_create", "srz_comm_destroy", "srz_frameset_exchange_bytes", "srz_frameset_allgather",
           "srz_frameset_deinterleave", "srz_frameset_allgather_inplace", "srz_frameset_gathered_row_offset",
           "srz_frameset_read_gathered_frame"]


class SrzError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"srz error {code}: {msg}")
        self.code = code


def lib():
    """Load libsrz.so (fails loudly if it was not built: run `python -c 'import __graft_entry__ as g; g.build()'`)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(f"{LIB_PATH} not found — the HIP extension is not built and there is no CPU fallback")
        L = C.CDLL(LIB_PATH)
        # a binding and a library that disagree on the ABI (a stale build, SRZ_LIB_PATH pointing at another checkout) must not
        # get as far as a call: argument MEANINGS change between versions (v3: the stream sentinel), not only signatures
        L.srz_abi_version.restype = C.c_int
        got = L.srz_abi_version()
        if got != abi.SRZ_ABI_VERSION:
            raise ImportError(f"{LIB_PATH} reports SRZ_ABI_VERSION {got}, this binding is written against {abi.SRZ_ABI_VERSION}: "
                              "rebuild the library (python -c 'import __graft_entry__ as g; g.build()') or fix SRZ_LIB_PATH")
        fp, vp = C.POINTER(C.c_float), C.c_void_p
        L.srz_create.argtypes = [C.POINTER(vp), C.c_int]
        L.srz_destroy.argtypes = [vp]
        L.srz_destroy.restype = None
        L.srz_last_error.argtypes = [vp]
        L.srz_last_error.restype = C.c_char_p
        L.srz_set_shard.argtypes = [vp, C.c_int, C.c_int]
        L.srz_set_option.argtypes = [vp, C.c_int, C.c_int]
        L.srz_texture_upload.argtypes = [vp, C.c_int, vp, C.c_int, C.c_int, C.c_int]
        L.srz_draw.argtypes = [vp, C.c_int, C.POINTER(abi.SrzFrame), fp, fp, fp, fp, C.POINTER(abi.SrzStats)]
        L.srz_draw_batch.argtypes = [vp, C.c_int, C.POINTER(abi.SrzFrame), C.c_int, C.POINTER(fp), C.POINTER(abi.SrzStats)]
        L.srz_frameset_create.argtypes = [vp, C.POINTER(abi.SrzFrame), C.c_int, C.POINTER(vp)]
        L.srz_sceneset_create.argtypes = [vp, C.POINTER(abi.SrzSceneFrame), C.c_int, C.POINTER(vp)]
        L.srz_mesh_upload.argtypes = [vp, C.c_int, vp, C.c_uint32, vp, C.c_uint32]
        L.srz_draw_scene.argtypes = [vp, C.c_int, C.POINTER(abi.SrzSceneFrame), fp, fp, fp, fp, C.POINTER(abi.SrzStats)]
        L.srz_frameset_destroy.argtypes = [vp, vp]
        L.srz_frameset_destroy.restype = None
        L.srz_frameset_local_rows.argtypes = [vp, vp]
        L.srz_frameset_out_bytes.argtypes = [vp, vp]
        L.srz_frameset_out_bytes.restype = C.c_size_t
        L.srz_frameset_render.argtypes = [vp, vp, vp, C.c_size_t, C.c_uint32, vp]
        L.srz_frameset_resolve8.argtypes = [vp, vp, vp, vp, C.c_size_t, vp]
        L.srz_frameset_stats.argtypes = [vp, vp, C.POINTER(abi.SrzStats)]
        L.srz_frameset_algorithmic_bytes.argtypes = [vp, vp]
        L.srz_frameset_algorithmic_bytes.restype = C.c_uint64
        L.srz_kernel_time_ms.argtypes = [vp, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_int)]
        L.srz_set_kernel_timing.argtypes = [vp, C.c_int]
        L.srz_kernel_time_samples.argtypes = [vp, C.POINTER(C.c_float), C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_double)]
        L.srz_sync.argtypes = [vp]
        L.srz_debug_counters.argtypes = [vp, C.POINTER(C.c_uint64), C.c_int]
        L.srz_verify_fastmath.argtypes = [vp, C.POINTER(C.c_uint64)]
        L.srz_verify_fastdiv.argtypes = [vp, C.POINTER(C.c_uint64)]
        L.srz_verify_fastpow.argtypes = [vp, C.c_float, C.POINTER(C.c_uint64)]
        L.srz_verify_fastlen.argtypes = [vp, C.POINTER(C.c_uint64)]
        L.srz_host_register.argtypes = [vp, C.c_void_p, C.c_size_t]
        L.srz_host_unregister.argtypes = [vp, C.c_void_p]
        L.srz_frameset_debug_counters.argtypes = [vp, vp, C.POINTER(C.c_uint32)]
        L.srz_comm_unique_id.argtypes = [vp]
        L.srz_comm_create.argtypes = [vp, vp, C.c_int, C.c_int, C.POINTER(vp)]
        L.srz_comm_destroy.argtypes = [vp, vp]
        L.srz_comm_destroy.restype = None
        L.srz_frameset_exchange_bytes.argtypes = [vp, vp, C.c_int]
        L.srz_frameset_exchange_bytes.restype = C.c_size_t
        L.srz_frameset_allgather.argtypes = [vp, vp, vp, vp, vp, vp, C.c_int, vp]
        L.srz_frameset_deinterleave.argtypes = [vp, vp, vp, vp, C.c_int, vp]
        L.srz_frameset_allgather_inplace.argtypes = [vp, vp, vp, vp, C.c_int, vp]
        L.srz_frameset_gathered_row_offset.argtypes = [vp, vp, C.c_int, C.c_int, C.c_int, C.c_int]
        L.srz_frameset_gathered_row_offset.restype = C.c_size_t
        L.srz_frameset_read_gathered_frame.argtypes = [vp, vp, vp, C.c_int, C.c_int, vp, vp]
        _lib = L
    return _lib


def _stream(stream):
    """hipStream_t argument of the C ABI: None → NULL = the ctx's own non-blocking stream; 0 (HIP's null stream, e.g. torch's
    default stream) → SRZ_STREAM_NULL, because work on the ctx's stream is NOT ordered against the null stream."""
    if stream is None:
        return None
    return C.c_void_p(-1 if stream == 0 else stream)


def _fp(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


class FrameSet:
    """Frames resident in HBM (srz_frameset_*)."""

    def __init__(self, ctx, frames):
        self.ctx, self.frames = ctx, list(frames)
        self.h = C.c_void_p()
        if isinstance(self.frames[0], abi.SceneFrame):  # meshes + matrices: vertex stage runs on the device
            arr = abi.scene_frames_array(self.frames)
            ctx._check(lib().srz_sceneset_create(ctx.h, arr, len(self.frames), C.byref(self.h)))
        else:
            arr = abi.frames_array(self.frames)
            ctx._check(lib().srz_frameset_create(ctx.h, arr, len(self.frames), C.byref(self.h)))
        self.n_frames = len(self.frames)
        self.width, self.height = self.frames[0].width, self.frames[0].height
        self.local_rows = lib().srz_frameset_local_rows(ctx.h, self.h)
        self.out_bytes = lib().srz_frameset_out_bytes(ctx.h, self.h)

    @property
    def out_shape(self):
        return (self.n_frames, 4, self.local_rows, self.width)

    def render(self, d_out_ptr, out_bytes, flags=abi.FUSED_CLEAR, stream=None):
        """d_out_ptr: integer device address (e.g. torch_tensor.data_ptr()). Asynchronous."""
        self.ctx._check(lib().srz_frameset_render(self.ctx.h, self.h, C.c_void_p(d_out_ptr), out_bytes, flags, _stream(stream)))

    def resolve8(self, d_planes_ptr, d_bgr8_ptr, bgr8_bytes, stream=None):
        """display()'s 8-bit resolve on the device: planes (render output) → [frame][rows][W][3] uint8."""
        self.ctx._check(lib().srz_frameset_resolve8(self.ctx.h, self.h, C.c_void_p(d_planes_ptr), C.c_void_p(d_bgr8_ptr), bgr8_bytes,
                                                    _stream(stream)))

    def debug_counters(self):
        """(tests) what the last render left: {slow_tiles, redo_tiles, pool_sub_cap, pool_demand, clear_wgs, clear_tuned}; waits for the device"""
        out = (C.c_uint32 * 6)()
        self.ctx._check(lib().srz_frameset_debug_counters(self.ctx.h, self.h, out))
        return dict(zip(("slow_tiles", "redo_tiles", "pool_sub_cap", "pool_demand", "clear_wgs", "clear_tuned"), (int(x) for x in out)))

    def exchange_bytes(self, what=abi.EXCHANGE_PLANES):
        return int(lib().srz_frameset_exchange_bytes(self.ctx.h, self.h, what))

    def allgather(self, comm, d_shard_ptr, d_gathered_ptr, d_full_ptr, what=abi.EXCHANGE_PLANES, stream=None):
        """RCCL all-gather of this rank's shard + de-interleave into row-major frames (srz_frameset_allgather)."""
        self.ctx._check(lib().srz_frameset_allgather(self.ctx.h, comm.h, self.h, C.c_void_p(d_shard_ptr), C.c_void_p(d_gathered_ptr),
                                                     C.c_void_p(d_full_ptr), what, _stream(stream)))

    def allgather_inplace(self, comm, d_gathered_ptr, what=abi.EXCHANGE_PLANES, stream=None):
        """the exchange without a second pass: this rank's shard was rendered at d_gathered + rank * exchange_bytes; ONE in-place
        RCCL all-gather fills in the others.  Layout: [rank][frame][plane][bands_per_rank*32][row] (gathered_row_offset)"""
        self.ctx._check(lib().srz_frameset_allgather_inplace(self.ctx.h, comm.h, self.h, C.c_void_p(d_gathered_ptr), what, _stream(stream)))

    def gathered_row_offset(self, frame, plane, row, what=abi.EXCHANGE_PLANES):
        """byte offset of row `row` of (frame, plane) in a rank-major gathered buffer; IndexError for a frame / plane / row /
        exchange kind the set does not have (the C call returns (size_t)-1 there: never add THAT to a device pointer)"""
        off = int(lib().srz_frameset_gathered_row_offset(self.ctx.h, self.h, what, frame, plane, row))
        if off == 2 ** 64 - 1:
            raise IndexError(f"gathered_row_offset: frame {frame} / plane {plane} / row {row} / kind {what} out of range")
        return off

    def read_gathered_frame(self, d_gathered_ptr, frame, what=abi.EXCHANGE_PLANES, stream=None):
        """one frame of a gathered buffer as row-major host planes ([4,H,W] float32, or [H,W,3] uint8), de-interleaved by the
        device→host copies"""
        out = np.empty((4, self.height, self.width), np.float32) if what == abi.EXCHANGE_PLANES else np.empty((self.height, self.width, 3), np.uint8)
        self.ctx._check(lib().srz_frameset_read_gathered_frame(self.ctx.h, self.h, C.c_void_p(d_gathered_ptr), what, frame,
                                                                out.ctypes.data_as(C.c_void_p), _stream(stream)))
        return out

    def deinterleave(self, d_gathered_ptr, d_full_ptr, what=abi.EXCHANGE_PLANES, stream=None):
        self.ctx._check(lib().srz_frameset_deinterleave(self.ctx.h, self.h, C.c_void_p(d_gathered_ptr), C.c_void_p(d_full_ptr), what,
                                                        _stream(stream)))

    def stats(self):
        st = abi.SrzStats()
        self.ctx._check(lib().srz_frameset_stats(self.ctx.h, self.h, C.byref(st)))
        return st.as_dict()

    def algorithmic_bytes(self):
        return int(lib().srz_frameset_algorithmic_bytes(self.ctx.h, self.h))

    def close(self):
        if self.h:
            lib().srz_frameset_destroy(self.ctx.h, self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Comm:
    """RCCL communicator of the band exchange (srz_comm_*): rank 0 makes the id, the host program distributes it."""

    def __init__(self, ctx, id128, rank, world):
        self.ctx, self.h = ctx, C.c_void_p()
        buf = (C.c_uint8 * 128).from_buffer_copy(bytes(id128))
        ctx._check(lib().srz_comm_create(ctx.h, buf, rank, world, C.byref(self.h)))
        self.rank, self.world = rank, world

    @staticmethod
    def unique_id():
        buf = (C.c_uint8 * 128)()
        rc = lib().srz_comm_unique_id(buf)
        if rc != 0:
            raise SrzError(rc, lib().srz_last_error(None).decode())
        return bytes(buf)

    def close(self):
        if self.h:
            lib().srz_comm_destroy(self.ctx.h, self.h)
            self.h = C.c_void_p()


class Context:
    """One ctx per process / GPU (srz_create)."""

    def __init__(self, device_id=0, rank=0, world=1):
        self.h = C.c_void_p()
        rc = lib().srz_create(C.byref(self.h), device_id)
        if rc != 0:
            raise SrzError(rc, lib().srz_last_error(None).decode())
        if world != 1:
            self.set_shard(rank, world)

    def _check(self, rc):
        if rc != 0:
            raise SrzError(rc, lib().srz_last_error(self.h).decode())

    def set_shard(self, rank, world):
        self._check(lib().srz_set_shard(self.h, rank, world))

    def set_option(self, option, value):
        """per-ctx switches for framesets created afterwards (abi.OPT_POOL_LAZY)"""
        self._check(lib().srz_set_option(self.h, option, int(value)))

    def texture_upload(self, tex_id, bgr):
        a = np.ascontiguousarray(bgr, dtype=np.uint8)
        h, w, c = a.shape
        assert c == 3
        self._check(lib().srz_texture_upload(self.h, tex_id, a.ctypes.data, w, h, w * 3))

    def mesh_upload(self, mesh_id, verts8, faces):
        """verts8: (nV,8) float32 [pos3 nrm3 uv2]; faces: (nF,3) uint32."""
        v = np.ascontiguousarray(verts8, dtype=np.float32)
        f = np.ascontiguousarray(faces, dtype=np.uint32)
        self._check(lib().srz_mesh_upload(self.h, mesh_id, v.ctypes.data, len(v), f.ctypes.data, len(f)))

    def draw(self, frame, planes=None, primitive=abi.PRIMITIVE_TRIANGLES, want_stats=False):
        """TraditionalRasterizer::draw for one scene; planes (z,c0,c1,c2) are modified in place."""
        if planes is None:
            h, w = frame.height, frame.width
            planes = (np.full((h, w), np.inf, np.float32), np.zeros((h, w), np.float32), np.zeros((h, w), np.float32),
                      np.zeros((h, w), np.float32))
        z, c0, c1, c2 = planes
        st = abi.SrzStats()
        fn = lib().srz_draw_scene if isinstance(frame, abi.SceneFrame) else lib().srz_draw
        self._check(fn(self.h, primitive, C.byref(frame.c), _fp(z), _fp(c0), _fp(c1), _fp(c2),
                       C.byref(st) if want_stats else None))
        return planes, (st.as_dict() if want_stats else None)

    def draw_batch(self, frames, planes=None, primitive=abi.PRIMITIVE_TRIANGLES, want_stats=False):
        """srz_draw_batch: planes = float32 array [n, 4, H, W] (z, c0, c1, c2 per frame), modified in place."""
        n, h, w = len(frames), frames[0].height, frames[0].width
        if planes is None:
            planes = np.zeros((n, 4, h, w), np.float32)
            planes[:, 0] = np.inf
        assert planes.dtype == np.float32 and planes.shape == (n, 4, h, w) and planes.flags.c_contiguous
        ptrs = (lib().srz_draw_batch.argtypes[4]._type_ * n)(*[_fp(planes[i]) for i in range(n)])
        st = abi.SrzStats()
        self._check(lib().srz_draw_batch(self.h, primitive, abi.frames_array(frames), n, ptrs, C.byref(st) if want_stats else None))
        return planes, (st.as_dict() if want_stats else None)

    def host_register(self, array):
        """page-lock a numpy array's memory: srz_draw / srz_draw_batch then move planes inside it by DMA (srz_host_register)"""
        self._check(lib().srz_host_register(self.h, array.ctypes.data, array.nbytes))

    def host_unregister(self, array):
        self._check(lib().srz_host_unregister(self.h, array.ctypes.data))

    def frameset(self, frames):
        return FrameSet(self, frames)

    def set_kernel_timing(self, on):
        """False/0: off; 1: whole launch set only (2 events per render); True/2: per-kernel groups as well (4 events)"""
        self._check(lib().srz_set_kernel_timing(self.h, 2 if on is True else int(on)))

    def kernel_time_ms(self, reset=True):
        ms, n = (C.c_double * 4)(), C.c_int()
        self._check(lib().srz_kernel_time_ms(self.h, 1 if reset else 0, ms, C.byref(n)))
        return {"bin_ms": ms[0], "raster_ms": ms[1], "shade_ms": ms[2], "total_ms": ms[3], "launches": n.value}

    def kernel_time_samples(self, cap=65536):
        """(per-render launch-set ms since the last reset, span ms from the first start to the last end); read it BEFORE
        kernel_time_ms(reset=True)"""
        out, n, span = (C.c_float * cap)(), C.c_int(), C.c_double()
        self._check(lib().srz_kernel_time_samples(self.h, out, cap, C.byref(n), C.byref(span)))
        return [float(out[i]) for i in range(n.value)], span.value

    def verify_fastmath(self):
        out = (C.c_uint64 * 4)()
        self._check(lib().srz_verify_fastmath(self.h, out))
        return [int(x) for x in out]

    def verify_fastdiv(self):
        out = (C.c_uint64 * 3)()
        self._check(lib().srz_verify_fastdiv(self.h, out))
        return [int(x) for x in out]

    def verify_fastpow(self, p):
        out = (C.c_uint64 * 4)()
        self._check(lib().srz_verify_fastpow(self.h, float(p), out))
        return [int(x) for x in out]

    def verify_fastlen(self):
        out = (C.c_uint64 * 5)()
        self._check(lib().srz_verify_fastlen(self.h, out))
        return [int(x) for x in out]

    def debug_counters(self):
        out = (C.c_uint64 * 32)()
        n = lib().srz_debug_counters(self.h, out, 32)
        return [int(out[i]) for i in range(n)]

    def sync(self):
        self._check(lib().srz_sync(self.h))

    def close(self):
        if self.h:
            lib().srz_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

"""ctypes binding of libsrz_host.so — the C++ host layer (SoftRasterizer::Scene & friends) that sits ABOVE the raster
boundary: OBJ / texture loading, model/view/projection matrices, vertex stage.  No GPU work happens here."""
import ctypes as C
import os

import numpy as np

from . import abi

_PKG = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB_PATH = os.path.join(_PKG, "libsrz_host.so")
_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(f"{LIB_PATH} not found — run __graft_entry__.build()")
        L = C.CDLL(LIB_PATH)
        fp, vp, cp = C.POINTER(C.c_float), C.c_void_p, C.c_char_p
        L.srzh_scene_create.argtypes = [cp, fp, fp, fp, C.c_int, C.c_int]
        L.srzh_scene_create.restype = vp
        L.srzh_scene_destroy.argtypes = [vp]
        L.srzh_scene_destroy.restype = None
        L.srzh_add_obj.argtypes = [vp, cp, cp, fp, C.c_float, fp, fp]
        L.srzh_add_shader.argtypes = [vp, cp, cp, vp, C.c_int, C.c_int, C.c_int]
        L.srzh_bind.argtypes = [vp, cp, cp]
        L.srzh_add_light.argtypes = [vp, cp, fp, fp]
        L.srzh_set_model.argtypes = [vp, cp, fp, C.c_float, fp, fp]
        L.srzh_set_view.argtypes = [vp, fp, fp, fp]
        L.srzh_set_view.restype = None
        L.srzh_set_projection.argtypes = [vp, C.c_float, C.c_float, C.c_float]
        L.srzh_set_projection.restype = None
        L.srzh_get_matrices.argtypes = [vp, fp, fp, fp]
        L.srzh_get_matrices.restype = None
        L.srzh_get_model.argtypes = [vp, cp, fp]
        L.srzh_mesh_counts.argtypes = [vp, cp, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
        L.srzh_mesh_copy.argtypes = [vp, cp, fp, vp]
        L.srzh_build_stream.argtypes = [vp]
        L.srzh_batch_info.argtypes = [vp, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_uint32), C.POINTER(vp),
                                      C.POINTER(C.c_int), C.POINTER(C.c_int)]
        L.srzh_batch_copy.argtypes = [vp, C.c_int, vp]
        L.srzh_n_mesh_draws.argtypes = [vp]
        L.srzh_mesh_draw.argtypes = [vp, C.c_int, C.c_char_p, C.POINTER(C.c_int), fp, fp, fp]
        L.srzh_n_lights.argtypes = [vp]
        L.srzh_lights_copy.argtypes = [vp, fp]
        L.srzh_set_reference_exact_lights.argtypes = [vp, C.c_int]
        L.srzh_set_reference_exact_lights.restype = None
        L.srzh_get_shader_constants.argtypes = [fp, fp, fp]
        L.srzh_get_shader_constants.restype = None
        L.srzh_image_size.argtypes = [cp, C.POINTER(C.c_int), C.POINTER(C.c_int)]
        L.srzh_image_copy.argtypes = [cp, vp]
        _lib = L
    return _lib


def _f3(v):
    a = np.ascontiguousarray(np.asarray(v, np.float32).reshape(3))
    return a, a.ctypes.data_as(C.POINTER(C.c_float))


class Scene:
    """SoftRasterizer::Scene registered on a (host-only) pipeline of the given resolution."""

    def __init__(self, name, eye, center, up, width, height):
        self._keep = []
        self.width, self.height = width, height
        self.eye = tuple(float(x) for x in eye)
        (_, e), (_, c), (_, u) = _f3(eye), _f3(center), _f3(up)
        self.h = lib().srzh_scene_create(name.encode(), e, c, u, width, height)
        if not self.h:
            raise RuntimeError("srzh_scene_create failed")
        self._tex_ids = {}

    def _chk(self, rc, what):
        if rc != 0:
            raise RuntimeError(f"{what} failed (see the log on stderr)")

    def add_obj(self, path, name, axis=(0, 1, 0), angle=0.0, translation=(0, 0, 0), scale=(1, 1, 1)):
        (_, a), (_, t), (_, s) = _f3(axis), _f3(translation), _f3(scale)
        self._chk(lib().srzh_add_obj(self.h, path.encode(), name.encode(), a, angle, t, s), "addGraphicObj/startLoadingMesh")

    def add_shader(self, name, texture_path, shader_type):
        self._chk(lib().srzh_add_shader(self.h, name.encode(), texture_path.encode(), None, 0, 0, int(shader_type)), "addShader")

    def bind(self, mesh, shader):
        self._chk(lib().srzh_bind(self.h, mesh.encode(), shader.encode()), "bindShader2Mesh")

    def add_light(self, name, position, intensity):
        (_, p), (_, i) = _f3(position), _f3(intensity)
        lib().srzh_add_light(self.h, name.encode(), p, i)

    def set_model(self, mesh, axis, angle, translation, scale):
        (_, a), (_, t), (_, s) = _f3(axis), _f3(translation), _f3(scale)
        self._chk(lib().srzh_set_model(self.h, mesh.encode(), a, angle, t, s), "setModelMatrix")

    def set_view(self, eye, center, up):
        self.eye = tuple(float(x) for x in eye)
        (_, e), (_, c), (_, u) = _f3(eye), _f3(center), _f3(up)
        lib().srzh_set_view(self.h, e, c, u)

    def set_projection(self, fovy, znear, zfar):
        lib().srzh_set_projection(self.h, fovy, znear, zfar)

    def matrices(self):
        v, p, n = (np.empty(16, np.float32) for _ in range(3))
        fp = C.POINTER(C.c_float)
        lib().srzh_get_matrices(self.h, v.ctypes.data_as(fp), p.ctypes.data_as(fp), n.ctypes.data_as(fp))
        return v, p, n

    def model(self, mesh):
        m = np.empty(16, np.float32)
        self._chk(lib().srzh_get_model(self.h, mesh.encode(), m.ctypes.data_as(C.POINTER(C.c_float))), "getModel")
        return m

    def mesh(self, name):
        nv, nf = C.c_uint32(), C.c_uint32()
        self._chk(lib().srzh_mesh_counts(self.h, name.encode(), C.byref(nv), C.byref(nf)), "getMeshObj")
        v = np.empty((nv.value, 8), np.float32)
        f = np.empty((nf.value, 3), np.uint32)
        lib().srzh_mesh_copy(self.h, name.encode(), v.ctypes.data_as(C.POINTER(C.c_float)), f.ctypes.data)
        return v, f

    def mesh_draws(self):
        """[(mesh_name, shader_type, ndc_mvp[16], normal_m[16])] in registration order, plus (zscale, zoffset)."""
        out, zz = [], np.zeros(2, np.float32)
        fp = C.POINTER(C.c_float)
        for i in range(lib().srzh_n_mesh_draws(self.h)):
            name = C.create_string_buffer(64)
            sh = C.c_int()
            mvp, nm = np.empty(16, np.float32), np.empty(16, np.float32)
            self._chk(lib().srzh_mesh_draw(self.h, i, name, C.byref(sh), mvp.ctypes.data_as(fp), nm.ctypes.data_as(fp),
                                           zz.ctypes.data_as(fp)), "meshDraws")
            out.append((name.value.decode(), sh.value, mvp, nm))
        return out, (float(zz[0]), float(zz[1]))

    def lights(self):
        n = lib().srzh_n_lights(self.h)
        out = np.zeros((max(n, 1), 6), np.float32)
        lib().srzh_lights_copy(self.h, out.ctypes.data_as(C.POINTER(C.c_float)))
        return out[:n].copy()

    def stream(self):
        """Scene::loadTriangleStream() → list of (shader_type, texture(bgr ndarray or None), tris[TRI_DTYPE])."""
        nb = lib().srzh_build_stream(self.h)
        out = []
        for b in range(nb):
            sh, nt, tp, tw, th = C.c_int(), C.c_uint32(), C.c_void_p(), C.c_int(), C.c_int()
            lib().srzh_batch_info(self.h, b, C.byref(sh), C.byref(nt), C.byref(tp), C.byref(tw), C.byref(th))
            tris = np.empty(nt.value, abi.TRI_DTYPE)
            lib().srzh_batch_copy(self.h, b, tris.ctypes.data)
            tex = None
            if tp.value:
                key = tp.value
                if key not in self._tex_ids:
                    buf = (C.c_uint8 * (tw.value * th.value * 3)).from_address(tp.value)
                    self._tex_ids[key] = np.frombuffer(buf, np.uint8).reshape(th.value, tw.value, 3).copy()
                tex = self._tex_ids[key]
            out.append((sh.value, tex, tris))
        return out

    def close(self):
        if self.h:
            lib().srzh_scene_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def shader_constants():
    ka, ks, pk = (np.empty(3, np.float32) for _ in range(3))
    fp = C.POINTER(C.c_float)
    lib().srzh_get_shader_constants(ka.ctypes.data_as(fp), ks.ctypes.data_as(fp), pk.ctypes.data_as(fp))
    return tuple(ka), tuple(ks), float(pk[0]), float(pk[1]), float(pk[2])


def load_image_bgr(path):
    w, h = C.c_int(), C.c_int()
    if lib().srzh_image_size(path.encode(), C.byref(w), C.byref(h)) != 0:
        raise RuntimeError("Cannot open file: " + path)
    out = np.empty((h.value, w.value, 3), np.uint8)
    lib().srzh_image_copy(path.encode(), out.ctypes.data)
    return out

"""ctypes mirror of include/srz.h (the C ABI of the raster + fragment-shade stage).

Pure data definitions: no library is loaded here, so the oracle wrapper (tests only) and the
product binding (srz/__init__.py) can both build identical `srz_frame` inputs.
"""
import ctypes as C

import numpy as np

SRZ_ABI_VERSION = 7  # = include/srz.h; srz.lib() refuses a library that reports another one
SRZ_OK = 0
SRZ_E_INVALID, SRZ_E_NODEVICE, SRZ_E_NOMEM, SRZ_E_TEXTURE, SRZ_E_PRIMITIVE = -1, -2, -3, -4, -5
SHADER_NORMAL, SHADER_TEXTURE, SHADER_PHONG, SHADER_DISPLACEMENT, SHADER_BUMP = 0, 1, 2, 3, 4
PRIMITIVE_LINES, PRIMITIVE_TRIANGLES = 0, 1
EXACT_SPLIT, UNIFIED, FUSED_CLEAR, ORDERED_RASTER, NO_Z_READBACK = 0, 1, 2, 4, 8
EXCHANGE_PLANES, EXCHANGE_BGR8 = 0, 1
OPT_POOL_LAZY, OPT_APPROX_SHADE = 1, 2  # srz_set_option

# numpy view of srz_tri (96 B): pos[3][3], nrm[3][3], uv[3][2]
TRI_DTYPE = np.dtype([("pos", "<f4", (3, 3)), ("nrm", "<f4", (3, 3)), ("uv", "<f4", (3, 2))])
LIGHT_DTYPE = np.dtype([("pos", "<f4", (3,)), ("intensity", "<f4", (3,))])
VERTEX_DTYPE = np.dtype([("pos", "<f4", (3,)), ("nrm", "<f4", (3,)), ("uv", "<f4", (2,))])
assert TRI_DTYPE.itemsize == 96 and LIGHT_DTYPE.itemsize == 24 and VERTEX_DTYPE.itemsize == 32


class SrzBatch(C.Structure):
    _fields_ = [("shader", C.c_int32), ("tex_id", C.c_int32), ("n_tris", C.c_uint32), ("_pad", C.c_uint32),
                ("tris", C.c_void_p)]


class SrzFrame(C.Structure):
    _fields_ = [("width", C.c_int32), ("height", C.c_int32), ("eye", C.c_float * 3), ("ka", C.c_float * 3),
                ("ks", C.c_float * 3), ("p", C.c_float), ("kh", C.c_float), ("kn", C.c_float),
                ("n_lights", C.c_uint32), ("n_batches", C.c_uint32), ("lights", C.c_void_p),
                ("batches", C.c_void_p), ("flags", C.c_uint32), ("_pad", C.c_uint32)]


class SrzMeshDraw(C.Structure):
    _fields_ = [("mesh_id", C.c_int32), ("shader", C.c_int32), ("tex_id", C.c_int32), ("_pad", C.c_int32),
                ("ndc_mvp", C.c_float * 16), ("normal_m", C.c_float * 16)]


class SrzSceneFrame(C.Structure):
    _fields_ = [("width", C.c_int32), ("height", C.c_int32), ("eye", C.c_float * 3), ("ka", C.c_float * 3),
                ("ks", C.c_float * 3), ("p", C.c_float), ("kh", C.c_float), ("kn", C.c_float),
                ("zscale", C.c_float), ("zoffset", C.c_float), ("n_lights", C.c_uint32), ("n_draws", C.c_uint32),
                ("lights", C.c_void_p), ("draws", C.c_void_p), ("flags", C.c_uint32), ("_pad", C.c_uint32)]


class SrzStats(C.Structure):
    _fields_ = [("n_tris", C.c_uint64), ("n_culled", C.c_uint64), ("pixel_tests", C.c_uint64),
                ("fragments", C.c_uint64), ("shaded", C.c_uint64), ("visible", C.c_uint64),
                ("visible_textured", C.c_uint64)]

    def as_dict(self):
        return {k: int(getattr(self, k)) for k, _ in self._fields_}


# Shader statics of the reference (src/Shader.cpp:7-12)
DEFAULT_KA = (0.005, 0.005, 0.005)
DEFAULT_KS = (0.7937, 0.7937, 0.7937)
DEFAULT_P, DEFAULT_KH, DEFAULT_KN = 150.0, 0.2, 0.1


class Frame:
    """Python-side owner of one srz_frame and the numpy arrays it points into."""

    def __init__(self, width, height, eye, lights, batches, flags=0, ka=DEFAULT_KA, ks=DEFAULT_KS, p=DEFAULT_P,
                 kh=DEFAULT_KH, kn=DEFAULT_KN):
        """lights: array-like (n,2,3) or LIGHT_DTYPE; batches: list of (shader, tex_id, tris[TRI_DTYPE])."""
        self.lights = np.ascontiguousarray(np.asarray(lights, dtype=np.float32).reshape(-1, 6)).view(LIGHT_DTYPE).reshape(-1) \
            if not (isinstance(lights, np.ndarray) and lights.dtype == LIGHT_DTYPE) else np.ascontiguousarray(lights)
        self.tris = []
        self._batches = (SrzBatch * max(1, len(batches)))()
        for i, (shader, tex_id, tris) in enumerate(batches):
            t = np.ascontiguousarray(tris)
            if t.dtype != TRI_DTYPE:
                t = np.ascontiguousarray(np.asarray(t, dtype=np.float32).reshape(-1, 24)).view(TRI_DTYPE).reshape(-1)
            self.tris.append(t)
            self._batches[i] = SrzBatch(int(shader), int(tex_id), len(t), 0, t.ctypes.data if len(t) else None)
        f = SrzFrame()
        f.width, f.height = int(width), int(height)
        f.eye[:] = [float(x) for x in eye]
        f.ka[:] = [float(x) for x in ka]
        f.ks[:] = [float(x) for x in ks]
        f.p, f.kh, f.kn = float(p), float(kh), float(kn)
        f.n_lights, f.n_batches = len(self.lights), len(batches)
        f.lights = self.lights.ctypes.data if len(self.lights) else None
        f.batches = C.cast(self._batches, C.c_void_p).value
        f.flags = int(flags)
        self.c = f

    @property
    def width(self):
        return self.c.width

    @property
    def height(self):
        return self.c.height

    @property
    def n_tris(self):
        return sum(len(t) for t in self.tris)

    def with_flags(self, flags):
        self.c.flags = int(flags)
        return self


class SceneFrame:
    """Python-side owner of one srz_scene_frame (meshes by slot + the vertex-stage matrices)."""

    def __init__(self, width, height, eye, lights, draws, zscale, zoffset, flags=0, ka=DEFAULT_KA, ks=DEFAULT_KS,
                 p=DEFAULT_P, kh=DEFAULT_KH, kn=DEFAULT_KN):
        """draws: list of (mesh_id, shader, tex_id, ndc_mvp[16], normal_m[16])."""
        self.lights = np.ascontiguousarray(np.asarray(lights, dtype=np.float32).reshape(-1, 6)).view(LIGHT_DTYPE).reshape(-1)
        self._draws = (SrzMeshDraw * max(1, len(draws)))()
        for i, (mesh_id, shader, tex_id, mvp, nm) in enumerate(draws):
            d = self._draws[i]
            d.mesh_id, d.shader, d.tex_id = int(mesh_id), int(shader), int(tex_id)
            d.ndc_mvp[:] = [float(x) for x in np.asarray(mvp, np.float32).reshape(16)]
            d.normal_m[:] = [float(x) for x in np.asarray(nm, np.float32).reshape(16)]
        f = SrzSceneFrame()
        f.width, f.height = int(width), int(height)
        f.eye[:] = [float(x) for x in eye]
        f.ka[:] = [float(x) for x in ka]
        f.ks[:] = [float(x) for x in ks]
        f.p, f.kh, f.kn = float(p), float(kh), float(kn)
        f.zscale, f.zoffset = float(zscale), float(zoffset)
        f.n_lights, f.n_draws = len(self.lights), len(draws)
        f.lights = self.lights.ctypes.data if len(self.lights) else None
        f.draws = C.cast(self._draws, C.c_void_p).value
        f.flags = int(flags)
        self.c = f

    @property
    def width(self):
        return self.c.width

    @property
    def height(self):
        return self.c.height


def scene_frames_array(frames):
    arr = (SrzSceneFrame * len(frames))()
    for i, f in enumerate(frames):
        arr[i] = f.c
    return arr


def frames_array(frames):
    arr = (SrzFrame * len(frames))()
    for i, f in enumerate(frames):
        arr[i] = f.c
    return arr

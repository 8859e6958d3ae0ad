"""The five BASELINE.json configs built with the ORACLE's host math — test inputs.

Frame index → rotation angle deg = 10*frame mod 360 (mirrors the ±10° steps of src/main.cpp:164-168).
"""
import functools
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "software-rasterizer_amd"))
from oracle import objload, oracle  # noqa: E402
from srz import abi  # noqa: E402

SPOT_OBJ = os.path.join(REPO, "assets/models/spot/spot_triangulated_good.obj")
SPOT_TEX = os.path.join(REPO, "assets/models/spot/spot_texture.png")
BUNNY_OBJ = os.path.join(REPO, "assets/models/bunny/bunny.obj")
CRATE_OBJ = os.path.join(REPO, "assets/models/Crate/Crate1.obj")
CRATE_TEX = os.path.join(REPO, "assets/models/Crate/Crate1.png")

# README.md:189-194
LIGHTS = np.array([[[0.9, 0.9, -0.9], [100, 100, 100]], [[0.0, 0.8, 0.9], [50, 50, 50]]], np.float32)
EYE = (0.0, 0.0, 0.9)
TEX_SPOT, TEX_CRATE = 0, 1


@functools.lru_cache(maxsize=None)
def mesh(path):
    v, _, f = objload.load_obj(path)
    return v, f


@functools.lru_cache(maxsize=None)
def spot_texture():
    return objload.load_texture_bgr(SPOT_TEX)


@functools.lru_cache(maxsize=None)
def crate_texture():
    return objload.load_texture_bgr(CRATE_TEX)


def camera(width, height, eye=EYE):
    """setViewMatrix + addScene(setNDCMatrix) + setProjectionMatrix(45.0f, 0.1f, 100.0f) (src/main.cpp:150-159)."""
    view = oracle.look_at_lh(eye, (0, 0, 0), (0, 1, 0))
    aspect = np.float32(width) / np.float32(height)
    proj = oracle.perspective_lh_no(45.0, aspect, 0.1, 100.0)
    ndc = oracle.ndc_matrix(width, height)
    return view, proj, ndc


def mesh_stream(path, width, height, angle, translation, scale, eye=EYE):
    v, f = mesh(path)
    view, proj, ndc = camera(width, height, eye)
    model = oracle.model_matrix((0, 1, 0), angle, translation, (scale,) * 3)
    return oracle.vertex_stage(v, f, model, view, proj, ndc, 0.1, 100.0)


def config1(flags=abi.FUSED_CLEAR):
    """256x256 plumbing: flat NORMAL triangle + one behind (z=60) + one in front (z=40)."""
    def tri(a, b, c, z):
        t = np.zeros(1, abi.TRI_DTYPE)
        t["pos"][0] = [[a[0], a[1], z], [b[0], b[1], z], [c[0], c[1], z]]
        t["nrm"][0] = [[0, 0, -1]] * 3
        return t
    tris = np.concatenate([tri((128, 40), (40, 200), (216, 200), 50.0),
                           tri((100, 60), (60, 180), (180, 180), 60.0),
                           tri((150, 100), (110, 220), (230, 210), 40.0)])
    return abi.Frame(256, 256, (0.0, 0.0, 1.0), LIGHTS, [(abi.SHADER_NORMAL, -1, tris)], flags)


def config2(frame_idx=0, size=1024, flags=abi.FUSED_CLEAR, shader=abi.SHADER_TEXTURE):
    deg = float((10 * frame_idx) % 360)
    tris = mesh_stream(SPOT_OBJ, size, size, deg, (0, 0, 0), 0.3)
    return abi.Frame(size, size, EYE, LIGHTS, [(shader, TEX_SPOT, tris)], flags)


def config3(frame_idx=0, width=1920, height=1080, flags=abi.FUSED_CLEAR):
    deg = float((10 * frame_idx) % 360)
    spot = mesh_stream(SPOT_OBJ, width, height, deg, (-0.25, 0, 0), 0.3)
    bunny = mesh_stream(BUNNY_OBJ, width, height, deg, (0.3, -0.2, 0), 2.0)
    return abi.Frame(width, height, EYE, LIGHTS, [(abi.SHADER_PHONG, -1, spot), (abi.SHADER_PHONG, -1, bunny)], flags)


def config4(frame_idx=0, size=2048, flags=abi.FUSED_CLEAR):
    deg = float((10 * frame_idx) % 360)
    batches = []
    for i in range(4):
        for j in range(4):
            t = (-0.375 + 0.25 * i, -0.375 + 0.25 * j, 0.0)
            batches.append((abi.SHADER_TEXTURE, TEX_SPOT, mesh_stream(SPOT_OBJ, size, size, deg, t, 0.1)))
    return abi.Frame(size, size, EYE, LIGHTS, batches, flags)


def config5(frame_idx=0, size=4096, flags=abi.FUSED_CLEAR):
    """8 depth-stacked spots submitted far-to-near, NORMAL / PHONG alternating."""
    deg = float((10 * frame_idx) % 360)
    batches = []
    for k in range(8):  # eye is at +z looking toward the origin: smaller z = farther
        sh = abi.SHADER_NORMAL if k % 2 == 0 else abi.SHADER_PHONG
        batches.append((sh, -1, mesh_stream(SPOT_OBJ, size, size, deg, (0, 0, 0.05 * k), 0.3)))
    return abi.Frame(size, size, EYE, LIGHTS, batches, flags)


README_EYE = (0.0, 0.0, -0.9)


def readme_scene(frame_idx=0, size=1024, flags=abi.FUSED_CLEAR):
    """The scene behind the reference's published raster timing (README.md:619-642; placement src/main.cpp:119-132, eye :150):
    spot + Crate1.obj (6 quads, fan-triangulated to 12 triangles), both TEXTURE with their own images, 1024x1024."""
    deg = float((10 * frame_idx) % 360)
    spot = mesh_stream(SPOT_OBJ, size, size, deg, (0.28, 0.1, 0.20), 0.2, README_EYE)
    crate = mesh_stream(CRATE_OBJ, size, size, deg, (0.28, -0.13, 0.15), 0.1, README_EYE)
    return abi.Frame(size, size, README_EYE, LIGHTS, [(abi.SHADER_TEXTURE, TEX_SPOT, spot), (abi.SHADER_TEXTURE, TEX_CRATE, crate)], flags)

"""The five BASELINE.json workloads, built with the PRODUCT's host layer (libsrz_host.so).

Frame index → rotation angle deg = 10*frame mod 360 (the ±10° steps of the reference's main loop, src/main.cpp:164-168);
camera / projection / lights / model placement are the README's (README.md:142-198) and src/main.cpp:150-159.
"""
import os

import numpy as np

from . import abi, host

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
SPOT_OBJ = os.path.join(REPO, "assets/models/spot/spot_triangulated_good.obj")
SPOT_TEX = os.path.join(REPO, "assets/models/spot/spot_texture.png")
BUNNY_OBJ = os.path.join(REPO, "assets/models/bunny/bunny.obj")
CRATE_OBJ = os.path.join(REPO, "assets/models/Crate/Crate1.obj")
CRATE_TEX = os.path.join(REPO, "assets/models/Crate/Crate1.png")

EYE = (0.0, 0.0, 0.9)
L1 = ((0.9, 0.9, -0.9), (100, 100, 100))
L2 = ((0.0, 0.8, 0.9), (50, 50, 50))
L3 = ((-0.9, 0.3, 0.6), (30, 30, 30))  # (a third light for the light-count variants of config 2; not in the README)
Y = (0, 1, 0)


class Workload:
    """A host Scene plus the per-frame model-matrix recipe; frame(i) returns an abi.Frame (post-MVP stream)."""

    def __init__(self, name, width, height, meshes, eye=EYE, lights=(L1, L2), p=None):
        """meshes: list of (mesh_name, obj_path, shader_type, translation, scale[, texture_path]); lights: ((pos, intensity), ...);
        p: Shader::p for these frames (None: the host layer's static, 150 as the reference ships it)."""
        meshes = [tuple(m) + ((SPOT_TEX,) if len(m) == 5 else ()) for m in meshes]
        self.name, self.width, self.height, self.eye = name, width, height, eye
        self.meshes = [m[:5] for m in meshes]
        sc = host.Scene(name, eye, (0, 0, 0), Y, width, height)
        textured = {}
        for (mname, path, shader, t, s, tex) in meshes:
            sc.add_obj(path, mname, Y, 0.0, t, (s, s, s))
            sname = f"shader{int(shader)}_{os.path.basename(tex)}"
            if sname not in textured:
                # every Shader needs a loadable image in the reference, NORMAL / PHONG included (src/Scene.cpp:158)
                sc.add_shader(sname, tex, shader)
                textured[sname] = True
            sc.bind(mname, sname)
        for i, l in enumerate(lights):
            sc.add_light(f"Light{i + 1}", *l)
        self.scene, self.p = sc, p
        self.textures = {}  # id(ndarray) -> slot
        self.texture_arrays = []
        self.mesh_tex_slot = {}  # mesh name -> texture slot (scene_frame)

    def _slot(self, tex):
        if tex is None:
            return -1
        k = id(tex)
        if k not in self.textures:
            self.textures[k] = len(self.texture_arrays)
            self.texture_arrays.append(tex)
        return self.textures[k]

    def frame(self, frame_idx, flags=abi.FUSED_CLEAR):
        deg = float((10 * frame_idx) % 360)
        sc = self.scene
        for (mname, _, _, t, s) in self.meshes:
            sc.set_model(mname, Y, deg, t, (s, s, s))
        sc.set_view(self.eye, (0, 0, 0), Y)
        sc.set_projection(45.0, 0.1, 100.0)  # raw 45 into a radians API, as the reference does (src/main.cpp:156-159)
        batches = []
        for (mesh, (shader, tex, tris)) in zip(self.meshes, sc.stream()):
            needs = shader in (abi.SHADER_TEXTURE, abi.SHADER_DISPLACEMENT, abi.SHADER_BUMP)
            batches.append((shader, self._slot(tex) if needs else -1, tris))
            if needs:
                self.mesh_tex_slot[mesh[0]] = batches[-1][1]
        ka, ks, p, kh, kn = host.shader_constants()
        return abi.Frame(self.width, self.height, sc.eye, sc.lights().reshape(-1, 2, 3), batches, flags, ka, ks, p if self.p is None else self.p, kh, kn)

    def upload_meshes(self, ctx):
        """Meshes resident on the GPU for the device vertex stage: slot i = i-th registered mesh."""
        for i, (mname, _, _, _, _) in enumerate(self.meshes):
            v, f = self.scene.mesh(mname)
            ctx.mesh_upload(i, v, f)

    def scene_frame(self, frame_idx, flags=abi.FUSED_CLEAR):
        """Same frame as frame(i) but as meshes + vertex-stage matrices (the vertex stage then runs on the device)."""
        deg = float((10 * frame_idx) % 360)
        sc = self.scene
        for (mname, _, _, t, s) in self.meshes:
            sc.set_model(mname, Y, deg, t, (s, s, s))
        sc.set_view(self.eye, (0, 0, 0), Y)
        sc.set_projection(45.0, 0.1, 100.0)
        if not self.texture_arrays:
            self.frame(frame_idx)  # registers the texture slots
        draws, (zs, zo) = sc.mesh_draws()
        slot = {m[0]: i for i, m in enumerate(self.meshes)}
        d = []
        for (name, shader, mvp, nm) in draws:
            needs = shader in (abi.SHADER_TEXTURE, abi.SHADER_DISPLACEMENT, abi.SHADER_BUMP)
            d.append((slot[name], shader, self.mesh_tex_slot.get(name, 0) if needs else -1, mvp, nm))
        ka, ks, p, kh, kn = host.shader_constants()
        return abi.SceneFrame(self.width, self.height, sc.eye, sc.lights().reshape(-1, 2, 3), d, zs, zo, flags, ka, ks, p if self.p is None else self.p, kh, kn)

    def upload_textures(self, ctx):
        for slot, tex in enumerate(self.texture_arrays):
            ctx.texture_upload(slot, tex)


def spot_texture_1024(shader=abi.SHADER_TEXTURE, size=1024, name="spot_texture_1024", **kw):
    """configs[1]: spot_triangulated_good.obj, 1024x1024, TEXTURE shader + 2 point lights."""
    return Workload(name, size, size, [("spot", SPOT_OBJ, shader, (0, 0, 0), 0.3)], **kw)


# configs[1] away from the benchmark's own lights / exponent / shader: what the reference allows at run time (any number of
# lights, Shader::p a mutable static, five shader types: src/Shader.cpp:7-12,192-640) and bench.py therefore also times
def spot_texture_1024_3lights():
    return spot_texture_1024(name="spot_texture_1024_3lights", lights=(L1, L2, L3))


def spot_texture_1024_p32():
    return spot_texture_1024(name="spot_texture_1024_p32", p=32.0)


def spot_texture_1024_p7_5():
    return spot_texture_1024(name="spot_texture_1024_p7.5", p=7.5)  # (not an integer: the generic build of k_shade)


def spot_bump_1024():
    return spot_texture_1024(shader=abi.SHADER_BUMP, name="spot_bump_1024")


def spot_bunny_1080p():
    """configs[2]: spot + bunny, 1920x1080, Blinn-Phong, 2 lights."""
    return Workload("spot_bunny_phong_1080p", 1920, 1080,
                    [("spot", SPOT_OBJ, abi.SHADER_PHONG, (-0.25, 0, 0), 0.3),
                     ("bunny", BUNNY_OBJ, abi.SHADER_PHONG, (0.3, -0.2, 0), 2.0)])


def spot_grid16_2048():
    """configs[3]: spot x16 (4x4 grid, s=0.1), 2048x2048, TEXTURE."""
    meshes = [(f"spot_{i}{j}", SPOT_OBJ, abi.SHADER_TEXTURE, (-0.375 + 0.25 * i, -0.375 + 0.25 * j, 0.0), 0.1)
              for i in range(4) for j in range(4)]
    return Workload("spot_x16_texture_2048", 2048, 2048, meshes)


def spot_overdraw8_4096():
    """configs[4]: 8 depth-stacked spots submitted far-to-near, NORMAL / PHONG alternating, 4096x4096."""
    meshes = [(f"spot_{k}", SPOT_OBJ, abi.SHADER_NORMAL if k % 2 == 0 else abi.SHADER_PHONG, (0, 0, 0.05 * k), 0.3)
              for k in range(8)]
    return Workload("spot_x8_overdraw_4096", 4096, 4096, meshes)


def readme_spot_crate_1024():
    """The scene of the reference's only published raster number (README.md:619-642, src/main.cpp:78-132): 1024x1024, spot
    (5856 triangles) + Crate1.obj (6 quads → 12 triangles), both TEXTURE with their own images, eye (0,0,-0.9); the three
    spheres of that scene contribute no raster triangles.  Lights: the README's two (README.md:189-194)."""
    return Workload("readme_spot_crate_1024", 1024, 1024,
                    [("spot", SPOT_OBJ, abi.SHADER_TEXTURE, (0.28, 0.1, 0.20), 0.2, SPOT_TEX),
                     ("Crate", CRATE_OBJ, abi.SHADER_TEXTURE, (0.28, -0.13, 0.15), 0.1, CRATE_TEX)], eye=(0.0, 0.0, -0.9))


WORKLOADS = {"spot_texture_1024_3lights": spot_texture_1024_3lights, "spot_texture_1024_p32": spot_texture_1024_p32,
             "spot_texture_1024_p7.5": spot_texture_1024_p7_5, "spot_bump_1024": spot_bump_1024,
             "readme_spot_crate_1024": readme_spot_crate_1024, "spot_texture_1024": spot_texture_1024, "spot_bunny_phong_1080p": spot_bunny_1080p,
             "spot_x16_texture_2048": spot_grid16_2048, "spot_x8_overdraw_4096": spot_overdraw8_4096}

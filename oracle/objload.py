"""Asset-prep checker — TEST INFRASTRUCTURE ONLY.

Python restatement of the reference's OBJ → (vertices, faces) conversion, used to check the product's C++
loader (software-rasterizer_amd/host) and to build inputs for the oracle:
  ObjLoader::startLoadingFromFile + processingVertexData   src/ObjLoader.cpp:78-233
  Tools::calculateNormalWithWeight                          src/Tools.cpp:234-248
tinyobjloader (un-vendored submodule, version unknown) is restated from its documented behaviour:
`v/vt/vn` 1-based or negative-relative indices, polygons fan-triangulated (v0, v(k-1), v(k)),
missing vertex colours = 1.0, absent vt/vn index = -1.  PARITY UNPINNED (no reference fixtures).
"""
import numpy as np


def _fix(i, n):
    i = int(i)
    return i - 1 if i > 0 else n + i


def parse_obj(path):
    v, vt, vn, col, corners = [], [], [], [], []
    with open(path, "r") as fh:
        for line in fh:
            s = line.split()
            if not s:
                continue
            if s[0] == "v":
                v.append([float(s[1]), float(s[2]), float(s[3])])
                col.append([float(s[4]), float(s[5]), float(s[6])] if len(s) >= 7 else [1.0, 1.0, 1.0])
            elif s[0] == "vt":
                vt.append([float(s[1]), float(s[2]) if len(s) > 2 else 0.0])
            elif s[0] == "vn":
                vn.append([float(s[1]), float(s[2]), float(s[3])])
            elif s[0] == "f":
                idx = []
                for tok in s[1:]:
                    p = tok.split("/")
                    vi = _fix(p[0], len(v))
                    ti = _fix(p[1], len(vt)) if len(p) > 1 and p[1] != "" else -1
                    ni = _fix(p[2], len(vn)) if len(p) > 2 and p[2] != "" else -1
                    idx.append((vi, ti, ni))
                for k in range(2, len(idx)):  # fan triangulation
                    corners += [idx[0], idx[k - 1], idx[k]]
    return (np.asarray(v, np.float32).reshape(-1, 3), np.asarray(vt, np.float32).reshape(-1, 2),
            np.asarray(vn, np.float32).reshape(-1, 3), np.asarray(col, np.float32).reshape(-1, 3), corners)


def _normalize(x):
    x = np.asarray(x, np.float32)
    d = np.float32(x[0] * x[0]) + np.float32(x[1] * x[1]) + np.float32(x[2] * x[2])
    return (x * (np.float32(1.0) / np.sqrt(d, dtype=np.float32))).astype(np.float32)


def _normal_with_weight(pa, pb, pc):
    ab, ac = (pb - pa).astype(np.float32), (pc - pa).astype(np.float32)
    n = np.array([ab[1] * ac[2] - ac[1] * ab[2], ab[2] * ac[0] - ac[2] * ab[0], ab[0] * ac[1] - ac[0] * ab[1]], np.float32)
    length = np.sqrt(np.float32(n @ n), dtype=np.float32)
    s = length / (np.sqrt(np.float32(ab @ ab), dtype=np.float32) * np.sqrt(np.float32(ac @ ac), dtype=np.float32))
    if not (-1e-8 <= length <= 1e-8):
        n = n * (np.arcsin(s, dtype=np.float32) / length)
    return _normalize(n)


def load_obj(path):
    """→ verts (nV,8) float32 [pos3 nrm3 uv2], colors (nV,3), faces (nF,3) uint32 — dedup in first-seen order."""
    v, vt, vn, col, corners = parse_obj(path)
    uniq, verts, cols, indices = {}, [], [], []
    no_normal = True
    for (vi, ti, ni) in corners:
        pos = v[vi]
        c = col[vi]
        nrm = np.zeros(3, np.float32)
        uv = np.zeros(2, np.float32)
        if ni >= 0:
            no_normal = False
            nrm = _normalize(vn[ni])                       # glm::normalize (src/ObjLoader.cpp:141-145)
        if ti >= 0:
            uv = np.array([vt[ti][0], np.float32(1.0) - vt[ti][1]], np.float32)  # v → 1 - v (:150-152)
        key = (float(pos[0]), float(pos[1]), float(pos[2]), float(c[0]), float(c[1]), float(c[2]),
               float(nrm[0]), float(nrm[1]), float(nrm[2]), float(uv[0]), float(uv[1]))
        j = uniq.get(key)
        if j is None:
            j = uniq[key] = len(verts)
            verts.append(np.concatenate([pos, nrm, uv]).astype(np.float32))
            cols.append(c)
        indices.append(j)
    verts = np.asarray(verts, np.float32).reshape(-1, 8)
    faces = np.asarray(indices, np.uint32).reshape(-1, 3)
    if no_normal:  # (:181-188) — later faces overwrite earlier ones per vertex
        for a, b, c in faces:
            pa, pb, pc = verts[a, :3].copy(), verts[b, :3].copy(), verts[c, :3].copy()
            verts[a, 3:6] = _normal_with_weight(pa, pb, pc)
            verts[b, 3:6] = _normal_with_weight(pb, pc, pa)
            verts[c, 3:6] = _normal_with_weight(pc, pa, pb)
    return verts, np.asarray(cols, np.float32).reshape(-1, 3), faces


def load_texture_bgr(path):
    """cv::imread(path) default flags (src/TextureLoader.cpp:4): 8-bit, 3 channels, BGR order, alpha dropped."""
    from PIL import Image
    im = Image.open(path).convert("RGB")
    return np.ascontiguousarray(np.asarray(im, np.uint8)[:, :, ::-1])

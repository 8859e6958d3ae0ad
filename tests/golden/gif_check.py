#!/usr/bin/env python3
"""The one artefact of this path's OUTPUT that the reference repository holds: assets/phong_cow.gif (README.md:223-231), a
screen capture (1280x720, 612 frames) of the PHONG spot scene of README.md:127-231 running in the reference's own window.
It cannot pin the oracle bit for bit (a scaled, palette-quantised capture of a window of unknown size), but it is the only
evidence that does not pass through our reading of the source: it fixes the silhouette of the mesh under the reference's view /
projection / NDC chain, the sense of rotation, the grey level / channel symmetry of the Blinn-Phong result — and, through the
grey levels INSIDE the silhouette (highlights, shaded side), how the image is turned: the capture is the render of the source
as written ROTATED BY 180 DEGREES (both axes: the sign convention of an older vertex stage, w < 0 after projection, or a flip
at display time — the capture cannot tell those two apart), not flipped in y and not mirrored.

  python tests/golden/gif_check.py extract     (needs /root/reference: writes tests/golden/phong_cow/frame_*.png, the
                                                window area of a few frames — data, committed)
  python tests/golden/gif_check.py             (CPU only: renders the README scene through the ORACLE for every 5 degrees of
                                                rotation, compares with the committed frames, prints the report as JSON)

Comparison: both silhouettes are cropped to their bounding boxes and resampled to 96x96 (the capture's scale and the window's
aspect ratio are unknown; SURVEY.md §8a-a2: the reference stretches x by W/H), then IoU; the oracle's rotation angle is the
one that maximises it.  tests/test_golden.py asserts the thresholds."""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
FRAMES = (0, 150, 300, 450)          # frames of the GIF kept as fixtures
WIN = (642, 22)                       # top-left corner of the window's client area in the 1280x720 capture
N = 96


def extract():
    from PIL import Image
    im = Image.open("/root/reference/assets/phong_cow.gif")
    os.makedirs(os.path.join(HERE, "phong_cow"), exist_ok=True)
    for i in FRAMES:
        im.seek(i)
        a = np.array(im.convert("RGB"))[WIN[1]:, WIN[0]:]
        Image.fromarray(a).save(os.path.join(HERE, "phong_cow", f"frame_{i:03d}.png"), optimize=True)
        print("frame", i, a.shape)


def load_frames():
    from PIL import Image
    return {i: np.array(Image.open(os.path.join(HERE, "phong_cow", f"frame_{i:03d}.png")).convert("RGB")).astype(np.float32)
            for i in FRAMES}


def normalise(mask, grey=None):
    """crop to the bounding box, resample to N x N (nearest for the mask, box mean for grey)"""
    ys, xs = np.where(mask)
    y0, y1, x0, x1 = ys.min(), ys.max() + 1, xs.min(), xs.max() + 1
    yi = (y0 + (np.arange(N) + 0.5) * (y1 - y0) / N).astype(int)
    xi = (x0 + (np.arange(N) + 0.5) * (x1 - x0) / N).astype(int)
    m = mask[np.ix_(yi, xi)]
    g = grey[np.ix_(yi, xi)] if grey is not None else None
    return m, g, (x1 - x0) / (y1 - y0)


def iou(a, b):
    return float((a & b).sum()) / float((a | b).sum())


def _variant(tris, size, how):
    """the post-MVP stream as an OLDER build of the reference might have produced it (hypotheses about the capture):
    rot180 = x and y negated around the screen centre (w < 0 after projection, the GAMES101-style sign convention; the winding
    is unchanged); flipy = the y flip src/Scene.cpp:330's comment promises (flipy_w: with the winding restored, i.e. front
    faces drawn); mirrorx likewise.  The lighting then happens in THAT screen space, as the shaders mix screen-space positions
    with world-space lights."""
    t = tris.copy()
    if how == "rot180":
        t["pos"][:, :, 0] = size - t["pos"][:, :, 0]
        t["pos"][:, :, 1] = size - t["pos"][:, :, 1]
    elif how.startswith("flipy"):
        t["pos"][:, :, 1] = size - t["pos"][:, :, 1]
    elif how.startswith("mirrorx"):
        t["pos"][:, :, 0] = size - t["pos"][:, :, 0]
    if how.endswith("_w"):
        for k in ("pos", "nrm", "uv"):
            a = t[k][:, 1].copy()
            t[k][:, 1] = t[k][:, 2]
            t[k][:, 2] = a
    return t


VERTEX_HYPOTHESES = ("rot180", "flipy", "flipy_w", "mirrorx", "mirrorx_w")


def oracle_views(size=512, step=5, how="as_written"):
    """the README PHONG scene through the oracle for every `step` degrees: {deg: (mask, grey, bbox aspect, BGR mean)}"""
    sys.path.insert(0, REPO)
    sys.path.insert(0, os.path.join(REPO, "software-rasterizer_amd"))
    sys.path.insert(0, os.path.join(REPO, "tests"))
    from oracle import oracle
    import scenes
    from srz import abi
    oracle.texture_set(scenes.TEX_SPOT, scenes.spot_texture())
    out = {}
    for deg in range(0, 360, step):
        tris = scenes.mesh_stream(scenes.SPOT_OBJ, size, size, float(deg), (0, 0, 0), 0.3)
        if how != "as_written":
            tris = _variant(tris, float(size), how)
        f = abi.Frame(size, size, scenes.EYE, scenes.LIGHTS, [(abi.SHADER_PHONG, -1, tris)], abi.FUSED_CLEAR)
        rc, planes, _ = oracle.draw(f)
        assert rc == 0
        mask = np.isfinite(planes[0])
        grey = (planes[1] + planes[2] + planes[3]) / 3.0
        m, g, aspect = normalise(mask, grey)
        out[deg] = (m, g, aspect, [float(planes[k][mask].mean()) for k in (1, 2, 3)])
    return out


def _grey_corr(m, g, vm, vg):
    both = m & vm
    return float(np.corrcoef(g[both], vg[both])[0, 1]) if both.sum() > 100 else 0.0


def _score(m, g, views, fn):
    """best silhouette IoU over the rotation angle; among the angles within 0.03 of it, the best correlation of the grey levels
    inside the common silhouette (where the highlights and the shaded side are: what tells a flip from a rotation — the mesh
    is nearly mirror-symmetric, so the silhouette alone cannot) → (iou, angle of the best IoU, correlation, its angle)"""
    ious = {d: iou(m, fn(v[0])) for d, v in views.items()}
    d_best = max(ious, key=ious.get)
    corr, d_corr = max((_grey_corr(m, g, fn(v[0]), fn(v[1])), d) for d, v in views.items() if ious[d] >= ious[d_best] - 0.03)
    return ious[d_best], d_best, corr, d_corr


def report():
    frames = load_frames()
    views = oracle_views()
    rep = {"frames": {}, "note": "per frame of the capture and per hypothesis: IoU of the bounding-box-normalised silhouettes (96x96) maximised over "
           "the rotation angle (5 degree steps), and the correlation of the grey levels inside the common silhouette at the best of the "
           "angles within 0.03 of that IoU.  display_*: the oracle's image of the source AS WRITTEN, turned at display time; vertex_*: the "
           "post-MVP positions turned (an older vertex stage), lit in that screen space"}
    orient = {"as_rendered": lambda x: x, "upside_down": lambda x: x[::-1], "mirrored": lambda x: x[:, ::-1], "rotated_180": lambda x: x[::-1, ::-1]}
    vviews = {h: oracle_views(how=h) for h in VERTEX_HYPOTHESES}
    ident = orient["as_rendered"]
    for i, a in frames.items():
        grey = a.mean(2)
        mask = grey > 40.0
        m, g, aspect = normalise(mask, grey)
        hyp = {"display_" + o: _score(m, g, views, fn) for o, fn in orient.items()}
        hyp.update({"vertex_" + h: _score(m, g, vviews[h], ident) for h in VERTEX_HYPOTHESES})
        score = {o: hyp["display_" + o] for o in orient}
        top = max(v[0] for v in score.values())
        o_best = max((o for o in score if score[o][0] >= top - 0.01), key=lambda o: score[o][2])   # silhouette first, then the grey levels
        best = score[o_best][1]
        vm, vg, vaspect, bgr = views[best]
        rep["frames"][str(i)] = {
            "best_orientation": o_best, "best_angle_deg": best, "iou": score[o_best][0], "grey_corr": score[o_best][2],
            "hypotheses": {h: {"iou": v[0], "angle_deg": v[1], "grey_corr": v[2], "grey_corr_angle_deg": v[3]} for h, v in hyp.items()},
            "best_iou_per_orientation": {o: {"iou": v[0], "angle_deg": v[1]} for o, v in score.items()},
            "iou_second_best_angle_apart": max(iou(m, orient[o_best](views[d][0])) for d in views if min((d - best) % 360, (best - d) % 360) >= 45),
            "bbox_aspect_gif": aspect, "bbox_aspect_oracle_square_frame": vaspect, "implied_window_w_over_h": aspect / vaspect,
            "mean_rgb_gif_in_silhouette": [float(a[..., c][mask].mean()) for c in range(3)],
            "mean_bgr_oracle_in_silhouette": bgr}
    angles = [rep["frames"][str(i)]["best_angle_deg"] for i in FRAMES]
    rep["rotation_deg_between_fixture_frames"] = [(angles[k + 1] - angles[k]) % 360 for k in range(len(angles) - 1)]
    return rep


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "extract":
        extract()
    else:
        print(json.dumps(report(), indent=1))

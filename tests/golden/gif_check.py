#!/usr/bin/env python3
"""The one artefact of this path's OUTPUT that the reference repository holds: assets/phong_cow.gif (README.md:223-231), a
screen capture (1280x720, 612 frames) of the PHONG spot scene of README.md:127-231 running in the reference's own window.
It cannot pin the oracle bit for bit (a scaled, palette-quantised capture of a window of unknown size), but it is the only
evidence that does not pass through our reading of the source: it fixes the image ORIENTATION (no y flip, no mirror), the
silhouette of the mesh under the reference's view / projection / NDC chain, the sense of rotation, and the grey level /
channel symmetry of the Blinn-Phong result.

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


def oracle_views(size=512, step=5):
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
        f = abi.Frame(size, size, scenes.EYE, scenes.LIGHTS, [(abi.SHADER_PHONG, -1, tris)], abi.FUSED_CLEAR)
        rc, planes, _ = oracle.draw(f)
        assert rc == 0
        mask = np.isfinite(planes[0])
        grey = (planes[1] + planes[2] + planes[3]) / 3.0
        m, g, aspect = normalise(mask, grey)
        out[deg] = (m, g, aspect, [float(planes[k][mask].mean()) for k in (1, 2, 3)])
    return out


def report():
    frames = load_frames()
    views = oracle_views()
    rep = {"frames": {}, "note": "IoU of bounding-box-normalised silhouettes (96x96) of the GIF's window and of the oracle's render of the "
           "README PHONG scene, maximised over the rotation angle (5 degree steps), for the oracle's image as rendered and flipped"}
    orient = {"as_rendered": lambda x: x, "upside_down": lambda x: x[::-1], "mirrored": lambda x: x[:, ::-1], "rotated_180": lambda x: x[::-1, ::-1]}
    for i, a in frames.items():
        grey = a.mean(2)
        mask = grey > 40.0
        m, g, aspect = normalise(mask, grey)
        score = {o: max((iou(m, fn(views[d][0])), d) for d in views) for o, fn in orient.items()}   # per orientation: (best IoU, angle)
        # the silhouette alone cannot tell a mirror image from the opposite rotation angle (the mesh is nearly symmetric): among
        # the orientations whose best IoU is within 0.01 of the maximum, the brightest region decides
        hl_gif = np.unravel_index(np.argmax(_blur(g * m)), g.shape)

        def highlight(o):
            vm_, vg_ = orient[o](views[score[o][1]][0]), orient[o](views[score[o][1]][1])
            return np.unravel_index(np.argmax(_blur(vg_ * vm_)), vg_.shape)

        top = max(v[0] for v in score.values())
        o_best = min((o for o in score if score[o][0] >= top - 0.01), key=lambda o: np.hypot(*(np.subtract(highlight(o), hl_gif))))
        best = score[o_best][1]
        vm, vg, vaspect, bgr = views[best]
        vm, vg = orient[o_best](vm), orient[o_best](vg)
        hl_orc = highlight(o_best)
        rep["frames"][str(i)] = {
            "best_orientation": o_best, "best_angle_deg": best, "iou": score[o_best][0],
            "best_iou_per_orientation": {o: {"iou": v[0], "angle_deg": v[1]} for o, v in score.items()},
            "iou_second_best_angle_apart": max(iou(m, orient[o_best](views[d][0])) for d in views if min((d - best) % 360, (best - d) % 360) >= 45),
            "bbox_aspect_gif": aspect, "bbox_aspect_oracle_square_frame": vaspect, "implied_window_w_over_h": aspect / vaspect,
            "mean_rgb_gif_in_silhouette": [float(a[..., c][mask].mean()) for c in range(3)],
            "mean_bgr_oracle_in_silhouette": bgr,
            "brightest_region_gif_xy": [hl_gif[1] / N, hl_gif[0] / N], "brightest_region_oracle_xy": [hl_orc[1] / N, hl_orc[0] / N]}
    angles = [rep["frames"][str(i)]["best_angle_deg"] for i in FRAMES]
    rep["rotation_deg_between_fixture_frames"] = [(angles[k + 1] - angles[k]) % 360 for k in range(len(angles) - 1)]
    return rep


def _blur(x, k=9):
    c = np.cumsum(np.cumsum(np.pad(x, ((k, 0), (k, 0))), 0), 1)
    return c[k:, k:] - c[:-k, k:] - c[k:, :-k] + c[:-k, :-k]


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "extract":
        extract()
    else:
        print(json.dumps(report(), indent=1))

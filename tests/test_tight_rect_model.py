"""-m "not gpu": the tightened rectangles of k_raster / bucket_group (software-rasterizer_amd/csrc/srz_kernels.hip: tight_margin,
slab_extent, clip_range) restated in numpy float32 and checked against the ORACLE's coverage: every pixel the reference's
bounding-box walk sets (src/Rasterizer.cpp:199-236, both the 8-wide and the scalar-tail test) must lie inside the rectangle
the kernels would walk for the tile that holds it.  The GPU parity tests check the kernels; this checks the rule itself — margin,
guard, slab clips — on tens of thousands of shapes without a GPU."""
import numpy as np
import pytest

import scenes  # noqa: F401  (sys.path)
from srz import abi
from test_oracle_kat import frame

F = np.float32
TILE = 32


def tight_margin(p, area2, box):
    """(ok, m) as the device function: D = the vertices' extent, ok iff the clamped box (sx, sy, ex, ey) lies within
    [min - 1, max] of the vertices on both axes, D^2 <= 256 |area2| and 2^-5 <= D <= 2^20"""
    mnx, mxx, mny, mxy = F(p[:, 0].min()), F(p[:, 0].max()), F(p[:, 1].min()), F(p[:, 1].max())
    d = F(max(F(mxx - mnx), F(mxy - mny)))
    m = F(d * F(0.001953125) + F(0.015625))
    sx, sy, ex, ey = box
    near = F(F(sx) + F(1.0)) >= mnx and F(ex) <= mxx and F(F(sy) + F(1.0)) >= mny and F(ey) <= mxy
    ok = bool(near) and bool(F(d * d) <= F(F(256.0) * abs(F(area2)))) and bool(d <= F(1048576.0)) and bool(d >= F(0.03125))
    return ok, m


def slab_extent(pu, pw, lo, hi):
    """extent along u of triangle ∩ {lo <= w <= hi}: vertices inside + crossings of the two bounds (float32, like the device)"""
    mn, mx = F(np.inf), F(-np.inf)
    for i in range(3):
        if lo <= pw[i] <= hi:
            mn, mx = min(mn, pu[i]), max(mx, pu[i])
    for i in range(3):
        j = (i + 1) % 3
        with np.errstate(divide="ignore", invalid="ignore", over="ignore"):
            inv = F(1.0) / F(pw[i] - pw[j])
            du = F(pu[j] - pu[i])
            for bound in (lo, hi):
                dp, dq = F(pw[i] - bound), F(pw[j] - bound)
                if (dp < 0) != (dq < 0):
                    u = F(F(du * F(dp * inv)) + pu[i])  # (the device uses an fma: within the margin's slack)
                    if not np.isnan(u):
                        mn, mx = min(mn, u), max(mx, u)
    return F(mn), F(mx)


def clip_range(i0, i1, mn, mx, m):
    f0, f1 = F(i0), F(i1)
    with np.errstate(invalid="ignore", over="ignore"):
        a = min(max(F(np.ceil(F(mn - m))), f0), F(f1 + 1))
        b = max(min(F(np.floor(F(mx + m))), f1), F(f0 - 1))
    return int(a), int(b)


def walked_rect(p, tx, ty, W, H):
    """the rectangle k_raster walks for triangle p in tile (tx, ty), or None — absolute pixel coordinates, inclusive"""
    x, y = p[:, 0], p[:, 1]
    bsx, bex = int(np.clip(x.min(), 0, W - 1)), int(np.clip(x.max(), 0, W - 1))
    bsy, bey = int(np.clip(y.min(), 0, H - 1)), int(np.clip(y.max(), 0, H - 1))
    X0, X1 = max(bsx, tx * TILE), min(bex, min(tx * TILE + TILE, W) - 1)
    Y0, Y1 = max(bsy, ty * TILE), min(bey, min(ty * TILE + TILE, H) - 1)
    if X0 > X1 or Y0 > Y1:
        return None
    abx, aby, acx, acy = F(x[1] - x[0]), F(y[1] - y[0]), F(x[2] - x[0]), F(y[2] - y[0])
    area2 = F(F(abx * acy) - F(aby * acx))
    ok, m = tight_margin(p, area2, (bsx, bsy, bex, bey))
    if ok:
        mn, mx = slab_extent(x, y, F(F(Y0) - m), F(F(Y1) + m))
        X0, X1 = clip_range(X0, X1, mn, mx, m)
        mn, mx = slab_extent(y, x, F(F(X0) - m), F(F(X1) + m))
        Y0, Y1 = clip_range(Y0, Y1, mn, mx, m)
    if X0 > X1 or Y0 > Y1:
        return None
    return X0, X1, Y0, Y1


def shapes(rng, n, W, H):
    """the adversarial families of tests/test_gpu_raster_paths.py, one triangle each"""
    kind = rng.integers(0, 9, n)
    c = rng.uniform([0, 0], [W, H], (n, 2))
    ang = rng.uniform(0, 2 * np.pi, n)
    d = np.stack([np.cos(ang), np.sin(ang)], 1)
    nrm = np.stack([-d[:, 1], d[:, 0]], 1)
    L = rng.uniform(20, 1.5 * max(W, H), n)[:, None]
    xy = np.zeros((n, 3, 2))
    k = kind == 0
    xy[k] = np.stack([c, c + nrm * (10.0 ** rng.uniform(-3, 0, n))[:, None], c + d * L], 1)[k]
    k = kind == 1
    xy[k] = np.stack([c, c + d * L, c + d * L * rng.uniform(0, 1, (n, 1)) + nrm * (10.0 ** rng.uniform(-4, -0.3, n))[:, None]], 1)[k]
    k = kind == 2
    xy[k] = (c[:, None, :] + rng.normal(size=(n, 3, 2)) * (10.0 ** rng.uniform(2, 6, (n, 1, 1))))[k]
    k = kind == 3
    far = c[:, None, :] + rng.normal(size=(n, 3, 2)) * 100.0
    far[:, 0] += d * (10.0 ** rng.uniform(6.3, 8, n))[:, None]
    xy[k] = far[k]
    k = kind == 4
    xy[k] = (np.round(c)[:, None, :] + rng.uniform(-1, 1, (n, 3, 2)) * (10.0 ** rng.uniform(-3, 0.3, (n, 1, 1))))[k]
    k = kind == 5
    grid = rng.choice([1.0, 0.5, 32.0, 0.25], (n, 1, 1))
    xy[k] = (np.round((c[:, None, :] + rng.uniform(-1, 1, (n, 3, 2)) * rng.uniform(2, 100, (n, 1, 1))) / grid) * grid)[k]
    k = kind == 6
    xy[k] = (c[:, None, :] + rng.uniform(-1, 1, (n, 3, 2)) * rng.uniform(1, 120, (n, 1, 1)))[k]
    # 7 tiny triangles, D in [2^-12, 2^-4], at coordinates < 64 (fine float grid) around pixel centres, near the sliver limit
    #   D^2 = 256 |area2| (height = D / 128 .. D): where the margin's constant is tightest
    k = kind == 7
    Dt = (2.0 ** rng.uniform(-12, -4, n))[:, None]
    c7 = np.round(rng.uniform([0, 0], [min(W, 64), min(H, 64)], (n, 2))) + rng.uniform(-1, 1, (n, 2)) * Dt
    xy[k] = np.stack([c7, c7 + d * Dt, c7 + d * Dt * rng.uniform(0, 1, (n, 1)) + nrm * Dt * (2.0 ** rng.uniform(-7.5, 0, n))[:, None]], 1)[k]
    # 8 small or ulp-sized triangles 1e4 .. 1e7 pixels off screen on ONE axis (the clamped box is an edge column / row)
    k = kind == 8
    off = np.zeros((n, 2))
    axis = rng.integers(0, 2, n)
    off[np.arange(n), axis] = rng.choice([-1.0, 1.0], n) * 10.0 ** rng.uniform(4, 7, n)
    c8 = c + off
    D8 = (10.0 ** rng.uniform(-3, 2.5, n))[:, None, None]
    xy[k] = (c8[:, None, :] + rng.uniform(-1, 1, (n, 3, 2)) * D8)[k]
    return xy.astype(np.float32)


@pytest.mark.parametrize("seed,flags", [(0, abi.FUSED_CLEAR), (1, abi.FUSED_CLEAR), (2, abi.FUSED_CLEAR | abi.UNIFIED), (3, abi.FUSED_CLEAR)])
def test_every_covered_pixel_lies_in_the_walked_rectangle(orc, seed, flags):
    rng = np.random.default_rng(7000 + seed)
    W, H = [(96, 80), (70, 130), (128, 64), (97, 97)][seed]
    n = 4000
    xy = shapes(rng, n, W, H)
    t = np.zeros(1, abi.TRI_DTYPE)
    t["nrm"][:] = [0, 0, -1]
    checked = covered_total = tightened = 0
    for i in range(n):
        for wind in (0, 1):  # (one of the two windings survives the backface test)
            p = xy[i] if wind == 0 else xy[i][::-1]
            t["pos"][0, :, :2] = p
            t["pos"][0, :, 2] = 5.0
            rc, ref, _ = orc.draw(frame(t, W, H, flags=flags))
            assert rc == 0
            ys, xs = np.nonzero(np.isfinite(ref[0]))
            if len(ys) == 0:
                continue
            checked += 1
            covered_total += len(ys)
            for ty in range(ys.min() // TILE, ys.max() // TILE + 1):
                # bucket_group: the tile range of band ty = the x-extent inside the band's rows (one slab clip from the box)
                selb = ys // TILE == ty
                if selb.any():
                    pf = p.astype(np.float32)
                    sx, ex = int(np.clip(pf[:, 0].min(), 0, W - 1)), int(np.clip(pf[:, 0].max(), 0, W - 1))
                    sy, ey = int(np.clip(pf[:, 1].min(), 0, H - 1)), int(np.clip(pf[:, 1].max(), 0, H - 1))
                    a2 = F(F(F(pf[1, 0] - pf[0, 0]) * F(pf[2, 1] - pf[0, 1])) - F(F(pf[1, 1] - pf[0, 1]) * F(pf[2, 0] - pf[0, 0])))
                    okb, mb = tight_margin(pf, a2, (sx, sy, ex, ey))
                    if okb:
                        mn, mx = slab_extent(pf[:, 0], pf[:, 1], F(F(max(ty * TILE, sy)) - mb), F(F(min(ty * TILE + TILE - 1, ey)) + mb))
                        sx, ex = clip_range(sx, ex, mn, mx, mb)
                    assert sx <= ex and xs[selb].min() // TILE >= sx // TILE and xs[selb].max() // TILE <= ex // TILE, \
                        (seed, i, wind, "band", ty, (sx, ex), (int(xs[selb].min()), int(xs[selb].max())), p.tolist())
                for tx in range(xs.min() // TILE, xs.max() // TILE + 1):
                    sel = (ys // TILE == ty) & (xs // TILE == tx)
                    if not sel.any():
                        continue
                    r = walked_rect(p.astype(np.float32), tx, ty, W, H)
                    assert r is not None, (seed, i, wind, "a tile with covered pixels was dropped", p.tolist())
                    X0, X1, Y0, Y1 = r
                    assert xs[sel].min() >= X0 and xs[sel].max() <= X1 and ys[sel].min() >= Y0 and ys[sel].max() <= Y1, \
                        (seed, i, wind, (tx, ty), r, (int(xs[sel].min()), int(xs[sel].max()), int(ys[sel].min()), int(ys[sel].max())), p.tolist())
                    bx0, bx1 = max(int(np.clip(p[:, 0].min(), 0, W - 1)), tx * TILE), min(int(np.clip(p[:, 0].max(), 0, W - 1)), tx * TILE + TILE - 1)
                    tightened += (X0 > bx0) or (X1 < bx1)
    assert checked > 1000 and covered_total > 100_000 and tightened > 500  # (the test tests something)

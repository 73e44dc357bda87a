"""The launch plan of the pyramid kernel, checked on the CPU.

The kernel chains several pyramid levels inside one workgroup: every tile of a group's last
level owns a share of every level of the group and computes a slightly larger range (what the
next level's range reads).  The plan (slam_amd_debug_pyramid_plan) is pure host arithmetic, so it
can be executed here in numpy exactly as the kernel executes it -- regions staged per tile,
source indices clamped to the staged range, only owned pixels written -- and compared with the
oracle's pyramid."""
import numpy as np
import pytest

from slam_amd import synth


def _tables(capi, src, dst):
    off, c1 = capi.linear_exact_coeffs(src, dst)
    pad = 8
    off = np.concatenate([off, np.full(pad, off[-1], off.dtype)]).astype(np.int64)
    c1 = np.concatenate([c1, np.full(pad, c1[-1], c1.dtype)]).astype(np.int64)
    return off, c1


def _run_plan(capi, params, img):
    h, w = img.shape
    plan = capi.pyramid_plan(params, w, h)
    ex_levels = None
    ws = [w]; hs = [h]
    for l in range(1, params.nlevels):
        s = np.float32(1.0) / np.float32(np.float64(params.scale_factor) ** l)
        ws.append(int(np.rint(np.float32(w) * s))); hs.append(int(np.rint(np.float32(h) * s)))
    levels = [img.astype(np.int64)] + [np.full((hs[l], (ws[l] + 3) // 4 * 4), -1, np.int64) for l in range(1, params.nlevels)]
    written = [None] + [np.zeros_like(levels[l], dtype=np.int32) for l in range(1, params.nlevels)]
    assert plan[0]["first"] == 1 and plan[-1]["last"] == params.nlevels - 1
    for g, nxt in zip(plan, plan[1:] + [None]):
        if nxt is not None:
            assert nxt["first"] == g["last"] + 1
        first, last = g["first"], g["last"]
        slots = last - first + 2
        assert g["lds_bytes"] <= 64 * 1024
        tabs = {l: (_tables(capi, ws[l - 1], ws[l]), _tables(capi, hs[l - 1], hs[l])) for l in range(first, last + 1)}
        for tj in range(g["nty"]):
            for ti in range(g["ntx"]):
                sx, sy = g["spanx"][ti], g["spany"][tj]
                # stage the source region; positions past the image read as zero
                src = levels[first - 1]
                x0, x1, y0, y1 = sx[0][0], sx[0][1], sy[0][0], sy[0][1]
                assert x0 % 4 == 0 and x0 >= 0 and y0 >= 0
                buf = np.zeros((y1 - y0, x1 - x0), np.int64)
                vh = min(y1, hs[first - 1]) - y0; vw = min(x1, ws[first - 1]) - x0
                buf[:vh, :vw] = src[y0:y0 + vh, x0:x0 + vw]
                for k in range(1, slots):
                    l = first + k - 1
                    (xo, xc), (yo, yc) = tabs[l]
                    cx0, cx1, ox0, ox1 = [int(v) for v in sx[k]]
                    cy0, cy1, oy0, oy1 = [int(v) for v in sy[k]]
                    px0, px1, py0, py1 = int(sx[k - 1][0]), int(sx[k - 1][1]), int(sy[k - 1][0]), int(sy[k - 1][1])
                    assert cx0 % 4 == 0 and cx1 % 4 == 0 and ox0 % 4 == 0 and ox1 % 4 == 0
                    assert cx0 <= ox0 and ox1 <= cx1 and cy0 <= oy0 and oy1 <= cy1 or ox1 <= ox0 or oy1 <= oy0
                    xs = np.arange(cx0, cx1); ys = np.arange(cy0, cy1)
                    xi = np.clip(xo[xs], px0, px1 - 2) - px0; yi = np.clip(yo[ys], py0, py1 - 2) - py0
                    c1 = xc[xs][None, :]; c0 = 256 - c1
                    m1 = yc[ys][:, None]; m0 = 256 - m1
                    h0 = buf[yi][:, xi] * c0 + buf[yi][:, xi + 1] * c1
                    h1 = buf[yi + 1][:, xi] * c0 + buf[yi + 1][:, xi + 1] * c1
                    out = (h0 * m0 + h1 * m1 + 32768) >> 16
                    if ox1 > ox0 and oy1 > oy0:
                        levels[l][oy0:oy1, ox0:ox1] = out[oy0 - cy0:oy1 - cy0, ox0 - cx0:ox1 - cx0]
                        written[l][oy0:oy1, ox0:ox1] += 1
                    buf = out
    return levels, written, ws, hs


@pytest.mark.parametrize("w,h,kw", [(200, 150, dict()), (333, 251, dict(scale_factor=1.5, nlevels=5)),
                                    (640, 480, dict()), (257, 190, dict(scale_factor=2.0, nlevels=4)),
                                    (300, 280, dict(scale_factor=1.1, nlevels=14)), (97, 61, dict(nlevels=3))])
def test_pyramid_plan_executes_to_the_oracle_pyramid(oracle, w, h, kw):
    from slam_amd import capi
    capi.lib()
    img = synth.scene(w, h, seed=3)
    params = capi.orb_params(nfeatures=500, **kw)
    levels, written, ws, hs = _run_plan(capi, params, img)
    ref = oracle.build_pyramid(oracle.params(nfeatures=500, **kw), img)
    for l in range(1, params.nlevels):
        assert ref[l].shape == (hs[l], ws[l])
        assert (written[l] == 1).all(), "level %d: every stored pixel belongs to exactly one tile" % l
        assert np.array_equal(levels[l][:, :ws[l]], ref[l].astype(np.int64)), "level %d" % l

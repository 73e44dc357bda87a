"""Seeded synthetic inputs for tests and benchmarks (numpy only, no GPU code).

Scenes: a sum of value-noise octaves plus random filled rectangles and discs, so
that FAST finds far more corners than any quota; a second view of a scene is a
homography warp of the first (bilinear).  Descriptor/keypoint sets for the
matcher-only workloads are i.i.d. uniform.
"""
import numpy as np

KEYPOINT_DTYPE = np.dtype(
    [("x", "<f4"), ("y", "<f4"), ("size", "<f4"), ("angle", "<f4"),
     ("response", "<f4"), ("octave", "<i4"), ("class_id", "<i4")]
)
DMATCH_DTYPE = np.dtype(
    [("queryIdx", "<i4"), ("trainIdx", "<i4"), ("imgIdx", "<i4"), ("distance", "<f4")]
)


def _value_noise(rng, w, h, cell):
    gw, gh = w // cell + 2, h // cell + 2
    grid = rng.random((gh, gw)).astype(np.float32)
    ys = np.arange(h, dtype=np.float32) / cell
    xs = np.arange(w, dtype=np.float32) / cell
    y0 = ys.astype(np.int32); x0 = xs.astype(np.int32)
    fy = (ys - y0)[:, None]; fx = (xs - x0)[None, :]
    g00 = grid[y0][:, x0]; g01 = grid[y0][:, x0 + 1]
    g10 = grid[y0 + 1][:, x0]; g11 = grid[y0 + 1][:, x0 + 1]
    return (g00 * (1 - fx) + g01 * fx) * (1 - fy) + (g10 * (1 - fx) + g11 * fx) * fy


def scene(w, h, seed=1, n_shapes=None):
    """8-bit grey test scene of size w x h."""
    rng = np.random.default_rng(seed)
    img = np.zeros((h, w), np.float32)
    for cell, amp in ((64, 90.0), (16, 50.0), (4, 25.0)):
        img += amp * _value_noise(rng, w, h, cell)
    img += 40.0
    if n_shapes is None:
        n_shapes = max(50, int(1200 * (w * h) / (640 * 480)))
    yy, xx = None, None
    for i in range(n_shapes):
        cx = int(rng.integers(0, w)); cy = int(rng.integers(0, h))
        r = int(rng.integers(3, 20))
        val = float(rng.integers(0, 256))
        x0, x1 = max(cx - r, 0), min(cx + r + 1, w)
        y0, y1 = max(cy - r, 0), min(cy + r + 1, h)
        if i % 2 == 0:
            img[y0:y1, x0:x1] = val
        else:
            yy = np.arange(y0, y1)[:, None] - cy
            xx = np.arange(x0, x1)[None, :] - cx
            m = (yy * yy + xx * xx) <= r * r
            sub = img[y0:y1, x0:x1]
            sub[m] = val
    return np.clip(np.rint(img), 0, 255).astype(np.uint8)


def homography(angle_deg=5.0, tx=12.0, ty=-7.0, scale=1.03, cx=0.0, cy=0.0):
    """3x3 matrix mapping destination pixels to source pixels (rotation about (cx, cy))."""
    a = np.deg2rad(angle_deg)
    c, s = np.cos(a) * scale, np.sin(a) * scale
    H = np.array([[c, -s, cx - c * cx + s * cy + tx],
                  [s, c, cy - s * cx - c * cy + ty],
                  [0, 0, 1]], np.float64)
    return H


def warp(img, H):
    """Bilinear warp: out(x, y) = img(H @ (x, y, 1)); outside pixels replicate the edge."""
    h, w = img.shape
    ys, xs = np.mgrid[0:h, 0:w].astype(np.float64)
    d = H[2, 0] * xs + H[2, 1] * ys + H[2, 2]
    sx = (H[0, 0] * xs + H[0, 1] * ys + H[0, 2]) / d
    sy = (H[1, 0] * xs + H[1, 1] * ys + H[1, 2]) / d
    sx = np.clip(sx, 0, w - 1.001); sy = np.clip(sy, 0, h - 1.001)
    x0 = np.floor(sx).astype(np.int64); y0 = np.floor(sy).astype(np.int64)
    fx = sx - x0; fy = sy - y0
    f = img.astype(np.float64)
    v = (f[y0, x0] * (1 - fx) + f[y0, x0 + 1] * fx) * (1 - fy) + \
        (f[y0 + 1, x0] * (1 - fx) + f[y0 + 1, x0 + 1] * fx) * fy
    return np.clip(np.rint(v), 0, 255).astype(np.uint8)


def frame_pair(w, h, seed=1):
    a = scene(w, h, seed)
    b = warp(a, homography(3.0, 6.0, -4.0, 1.02, w / 2.0, h / 2.0))
    return a, b


def frame_sequence(w, h, n, seed=1):
    """n views of one scene under a slowly drifting homography."""
    base = scene(w, h, seed)
    out = []
    for i in range(n):
        H = homography(0.15 * i, 0.8 * i, -0.5 * i, 1.0 + 0.0005 * i, w / 2.0, h / 2.0)
        out.append(base if i == 0 else warp(base, H))
    return out


def random_descriptors(n, seed=3):
    rng = np.random.default_rng(seed)
    return rng.integers(0, 256, size=(n, 32), dtype=np.uint8)


def random_keypoints(n, w=640, h=480, nlevels=8, seed=4):
    rng = np.random.default_rng(seed)
    kp = np.zeros(n, KEYPOINT_DTYPE)
    kp["x"] = rng.random(n, dtype=np.float32) * w
    kp["y"] = rng.random(n, dtype=np.float32) * h
    kp["octave"] = rng.integers(0, nlevels, size=n)
    kp["size"] = 31.0
    kp["angle"] = rng.random(n, dtype=np.float32) * 360
    kp["response"] = rng.integers(20, 255, size=n).astype(np.float32)
    kp["class_id"] = -1
    return kp


def noisy_copy(desc, flip_bits=20, seed=5):
    """Copy of a descriptor set with `flip_bits` random bits flipped per row."""
    rng = np.random.default_rng(seed)
    out = desc.copy()
    n = len(desc)
    for _ in range(flip_bits):
        byte = rng.integers(0, 32, size=n); bit = rng.integers(0, 8, size=n)
        out[np.arange(n), byte] ^= (1 << bit).astype(np.uint8)
    return out


def two_view_setup(n, seed):
    """Two cameras looking down +z, points in front of both, exact keypoints = projections."""
    rng = np.random.default_rng(seed)
    K4 = np.array([520.0, 515.0, 320.0, 240.0])
    Rt_q = np.hstack([np.eye(3), np.zeros((3, 1))])
    ang = 0.05
    R = np.array([[np.cos(ang), 0, np.sin(ang)], [0, 1, 0], [-np.sin(ang), 0, np.cos(ang)]])
    t = np.array([[-0.4], [0.02], [0.01]])
    Rt_t = np.hstack([R, t])
    cq = np.zeros(3, np.float32)
    ct = (-R.T @ t).ravel().astype(np.float32)
    xyz = np.stack([rng.uniform(-1, 1, n), rng.uniform(-0.7, 0.7, n), rng.uniform(2, 6, n)], 1).astype(np.float32)
    return K4, Rt_q, Rt_t, cq, ct, xyz

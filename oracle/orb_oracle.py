"""ctypes binding of the CPU oracle (oracle/orb_oracle.c).

Test infrastructure only: imported by tests/, by the smoke check and by the CPU
baseline leg of bench.py, never by the shipped package.  Parity unpinned (see the
header of orb_oracle.h).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "liborb_oracle.so")

KEYPOINT_DTYPE = np.dtype(
    [("x", "<f4"), ("y", "<f4"), ("size", "<f4"), ("angle", "<f4"),
     ("response", "<f4"), ("octave", "<i4"), ("class_id", "<i4")]
)
DMATCH_DTYPE = np.dtype(
    [("queryIdx", "<i4"), ("trainIdx", "<i4"), ("imgIdx", "<i4"), ("distance", "<f4")]
)
assert KEYPOINT_DTYPE.itemsize == 28 and DMATCH_DTYPE.itemsize == 16

OK, ERR_ARG, ERR_UNSUPPORTED, ERR_CAPACITY, ERR_NOMEM = 0, -1, -2, -3, -4


class Params(C.Structure):
    _fields_ = [
        ("nfeatures", C.c_int32), ("scale_factor", C.c_float), ("nlevels", C.c_int32),
        ("edge_threshold", C.c_int32), ("first_level", C.c_int32), ("wta_k", C.c_int32),
        ("score_type", C.c_int32), ("patch_size", C.c_int32), ("fast_threshold", C.c_int32),
    ]


def build(force=False):
    """Compile the oracle with gcc if the shared library is missing or stale."""
    src = os.path.join(_HERE, "orb_oracle.c")
    deps = [src, os.path.join(_HERE, "orb_oracle.h"), os.path.join(_HERE, "..", "include", "orb_pattern_31.inc")]
    stale = force or not os.path.exists(_LIB_PATH) or any(
        os.path.getmtime(d) > os.path.getmtime(_LIB_PATH) for d in deps)
    if stale:
        subprocess.check_call(["make", "-C", _HERE, "-B", "liborb_oracle.so"],
                              stdout=subprocess.DEVNULL)
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        _lib = C.CDLL(_LIB_PATH)
        _lib.orc_fast_atan2.restype = C.c_float
        _lib.orc_fast_atan2.argtypes = [C.c_float, C.c_float]
        _lib.orc_level_scale.restype = C.c_float
        _lib.orc_level_scale.argtypes = [C.POINTER(Params), C.c_int]
    return _lib


def use_native_build():
    """Switches this module to a build of the same source made for the CPU it runs on
    (gcc -O3 -march=native -ffp-contract=off, in a temporary directory), for timing the CPU
    baseline.  Returns a description of the build in use; on any failure the -O2 library
    stays in use."""
    global _lib
    import tempfile
    flags = ["-O3", "-march=native", "-std=c99", "-fPIC", "-ffp-contract=off", "-fopenmp"]
    try:
        out = os.path.join(tempfile.mkdtemp(prefix="orb_oracle_native_"), "liborb_oracle_native.so")
        subprocess.check_call(["gcc"] + flags + ["-shared", "-o", out, os.path.join(_HERE, "orb_oracle.c"), "-lm"],
                              stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        new = C.CDLL(out)
    except (OSError, subprocess.CalledProcessError):
        lib()
        return "gcc -O2 (native build failed)"
    _lib = None
    lib()                      # sets the restype / argtypes on the default library object ...
    new.orc_fast_atan2.restype = C.c_float
    new.orc_fast_atan2.argtypes = [C.c_float, C.c_float]
    new.orc_level_scale.restype = C.c_float
    new.orc_level_scale.argtypes = [C.POINTER(Params), C.c_int]
    _lib = new                 # ... and the native one takes its place
    return "gcc " + " ".join(flags[:2] + flags[4:5])


def params(nfeatures=1000, scale_factor=1.2, nlevels=8, edge_threshold=31, fast_threshold=20,
           first_level=0, wta_k=2, score_type=1, patch_size=31):
    return Params(nfeatures, scale_factor, nlevels, edge_threshold, first_level, wta_k,
                  score_type, patch_size, fast_threshold)


def _p(a, t=C.c_void_p):
    return a.ctypes.data_as(t) if a is not None else None


def _img(img):
    img = np.ascontiguousarray(img, dtype=np.uint8)
    assert img.ndim == 2
    return img


def pyramid_sizes(p, w, h):
    ws = np.zeros(16, np.int32); hs = np.zeros(16, np.int32)
    rc = lib().orc_pyramid_sizes(C.byref(p), w, h, _p(ws), _p(hs))
    if rc: raise ValueError("orc_pyramid_sizes rc=%d" % rc)
    return ws[:p.nlevels].copy(), hs[:p.nlevels].copy()


def level_quotas(p):
    q = np.zeros(16, np.int32)
    rc = lib().orc_level_quotas(C.byref(p), _p(q))
    if rc: raise ValueError("orc_level_quotas rc=%d" % rc)
    return q[:p.nlevels].copy()


def level_scale(p, level):
    return float(lib().orc_level_scale(C.byref(p), level))


def linear_exact_coeffs(src, dst):
    off = np.zeros(dst, np.int32); c1 = np.zeros(dst, np.uint16)
    rc = lib().orc_linear_exact_coeffs(src, dst, _p(off), _p(c1))
    if rc: raise ValueError(rc)
    return off, c1


def resize_linear_exact(img, dw, dh):
    img = _img(img)
    out = np.zeros((dh, dw), np.uint8)
    rc = lib().orc_resize_linear_exact(_p(img), img.shape[1], img.shape[0], img.strides[0],
                                       _p(out), dw, dh, dw)
    if rc: raise ValueError(rc)
    return out


def build_pyramid(p, img):
    img = _img(img)
    h, w = img.shape
    ws, hs = pyramid_sizes(p, w, h)
    buf = np.zeros(int((ws.astype(np.int64) * hs).sum()), np.uint8)
    rc = lib().orc_build_pyramid(C.byref(p), _p(img), w, h, img.strides[0], _p(buf))
    if rc: raise ValueError(rc)
    levels, o = [], 0
    for lw, lh in zip(ws, hs):
        levels.append(buf[o:o + lw * lh].reshape(lh, lw))
        o += lw * lh
    return levels


def fast_score_map(img, threshold=20):
    img = _img(img)
    h, w = img.shape
    out = np.zeros((h, w), np.uint8)
    rc = lib().orc_fast_score_map(_p(img), w, h, img.strides[0], threshold, _p(out))
    if rc: raise ValueError(rc)
    return out


def fast_detect(img, threshold=20):
    img = _img(img)
    h, w = img.shape
    cap = w * h // 2 + 64
    xys = np.zeros((cap, 3), np.int32)
    n = C.c_int(0)
    rc = lib().orc_fast_detect(_p(img), w, h, img.strides[0], threshold, _p(xys), cap, C.byref(n))
    if rc: raise ValueError(rc)
    return xys[:n.value].copy()


def gaussian_blur7(img):
    img = _img(img)
    h, w = img.shape
    out = np.zeros((h, w), np.uint8)
    rc = lib().orc_gaussian_blur7(_p(img), w, h, img.strides[0], _p(out), w)
    if rc: raise ValueError(rc)
    return out


def fast_atan2(y, x):
    return float(lib().orc_fast_atan2(float(y), float(x)))


def ic_moments(img, x, y):
    img = _img(img)
    m01 = C.c_int(0); m10 = C.c_int(0)
    lib().orc_ic_moments(_p(img), img.strides[0], x, y, C.byref(m01), C.byref(m10))
    return m01.value, m10.value


def brief_descriptor(blurred, cx, cy, angle_deg):
    blurred = _img(blurred)
    d = np.zeros(32, np.uint8)
    lib().orc_brief_descriptor(_p(blurred), blurred.strides[0], cx, cy, C.c_float(angle_deg), _p(d))
    return d


def quadtree_select(cand, region, target):
    """cand: [n, 3] int32 (x, y, score); region = (x0, y0, x1, y1).  Returns the keep flags."""
    cand = np.ascontiguousarray(cand, np.int32).reshape(-1, 3)
    keep = np.zeros(len(cand), np.uint8)
    rc = lib().orc_quadtree_select(_p(cand), len(cand), int(region[0]), int(region[1]), int(region[2]), int(region[3]),
                                   int(target), _p(keep))
    if rc: raise ValueError("orc_quadtree_select rc=%d" % rc)
    return keep.astype(bool)


def detect_and_compute(p, img, capacity=None, grid_cell=0, quadtree=False):
    img = _img(img)
    h, w = img.shape
    if capacity is None:
        capacity = max(4 * p.nfeatures, 1024)
    while True:
        kps = np.zeros(capacity, KEYPOINT_DTYPE)
        desc = np.zeros((capacity, 32), np.uint8)
        n = C.c_int(0)
        rc = lib().orc_detect_and_compute_ex(C.byref(p), int(grid_cell), int(bool(quadtree)), _p(img), w, h,
                                             img.strides[0], _p(kps), _p(desc), capacity, C.byref(n))
        if rc == ERR_CAPACITY:
            capacity = n.value
            continue
        if rc: raise ValueError("orc_detect_and_compute rc=%d" % rc)
        return kps[:n.value].copy(), desc[:n.value].copy()


def max_threads():
    return int(lib().orc_max_threads())


def detect_and_compute_batch(p, frames, capacity=4096):
    """frames: [n, h, w] uint8.  Returns a list of (keypoints, descriptors); frames run in parallel."""
    frames = np.ascontiguousarray(frames, np.uint8)
    n, h, w = frames.shape
    kps = np.zeros((n, capacity), KEYPOINT_DTYPE)
    desc = np.zeros((n, capacity, 32), np.uint8)
    counts = np.zeros(n, np.int32)
    lib().orc_detect_and_compute_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_size_t, C.c_int, C.c_int,
                                                   C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
    rc = lib().orc_detect_and_compute_batch(C.cast(C.byref(p), C.c_void_p), _p(frames), n, h * w, w, h, w,
                                            _p(kps), _p(desc), capacity, _p(counts))
    if rc: raise ValueError("orc_detect_and_compute_batch rc=%d" % rc)
    return [(kps[f, :counts[f]].copy(), desc[f, :counts[f]].copy()) for f in range(n)]


def match_gated_mt(qd, qkp, td, tkp, max_dist=300.0, area=-1.0, max_level=4):
    return _match(lib().orc_match_gated_mt, qd, qkp, td, tkp, max_dist, area, max_level)


def hamming256(a, b):
    a = np.ascontiguousarray(a, np.uint8); b = np.ascontiguousarray(b, np.uint8)
    return int(lib().orc_hamming256(_p(a), _p(b)))


def _match(fn, qd, qkp, td, tkp, max_dist, area, max_level):
    qd = np.ascontiguousarray(qd, np.uint8).reshape(-1, 32)
    td = np.ascontiguousarray(td, np.uint8).reshape(-1, 32)
    qkp = np.ascontiguousarray(qkp, KEYPOINT_DTYPE); tkp = np.ascontiguousarray(tkp, KEYPOINT_DTYPE)
    nq, nt = len(qd), len(td)
    assert len(qkp) == nq and len(tkp) == nt
    out = np.zeros(max(nq, 1), DMATCH_DTYPE)
    n = C.c_int(0)
    rc = fn(_p(qd), _p(qkp), nq, _p(td), _p(tkp), nt, C.c_float(max_dist), C.c_float(area),
            int(max_level), _p(out), C.byref(n))
    if rc: raise ValueError(rc)
    return out[:n.value].copy()


def radius_match_filter(qd, qkp, td, tkp, max_dist=300.0, area=-1.0, max_level=4):
    return _match(lib().orc_radius_match_filter, qd, qkp, td, tkp, max_dist, area, max_level)


def match_gated(qd, qkp, td, tkp, max_dist=300.0, area=-1.0, max_level=4):
    return _match(lib().orc_match_gated, qd, qkp, td, tkp, max_dist, area, max_level)


def match_gated_ex(qd, qkp, td, tkp, max_dist=300.0, area=-1.0, max_level=4, ratio=0.0, histo_length=0):
    qd = np.ascontiguousarray(qd, np.uint8).reshape(-1, 32)
    td = np.ascontiguousarray(td, np.uint8).reshape(-1, 32)
    qkp = np.ascontiguousarray(qkp, KEYPOINT_DTYPE); tkp = np.ascontiguousarray(tkp, KEYPOINT_DTYPE)
    out = np.zeros(max(len(qd), 1), DMATCH_DTYPE)
    n = C.c_int(0)
    rc = lib().orc_match_gated_ex(_p(qd), _p(qkp), len(qd), _p(td), _p(tkp), len(td), C.c_float(max_dist),
                                  C.c_float(area), int(max_level), C.c_float(ratio), int(histo_length),
                                  _p(out), C.byref(n))
    if rc: raise ValueError(rc)
    return out[:n.value].copy()


def frame_match(d1, kp1, d2, kp2, ids=None, max_dist=300.0, area=-1.0, max_level=4):
    d1 = np.ascontiguousarray(d1, np.uint8).reshape(-1, 32)
    d2 = np.ascontiguousarray(d2, np.uint8).reshape(-1, 32)
    kp1 = np.ascontiguousarray(kp1, KEYPOINT_DTYPE); kp2 = np.ascontiguousarray(kp2, KEYPOINT_DTYPE)
    ids_a = None if ids is None or len(ids) == 0 else np.ascontiguousarray(ids, np.int32)
    nq = len(d1) if ids_a is None else len(ids_a)
    out = np.zeros(max(nq, 1), DMATCH_DTYPE)
    n = C.c_int(0)
    rc = lib().orc_frame_match(_p(d1), _p(kp1), len(d1), _p(d2), _p(kp2), len(d2),
                               _p(ids_a), 0 if ids_a is None else len(ids_a),
                               C.c_float(max_dist), C.c_float(area), int(max_level),
                               _p(out), C.byref(n))
    if rc: raise ValueError(rc)
    return out[:n.value].copy()


def project_points(Rt, K4, dist5, xyz):
    Rt = np.ascontiguousarray(Rt, np.float64).reshape(12)
    K4 = np.ascontiguousarray(K4, np.float64).reshape(4)
    d5 = None if dist5 is None else np.ascontiguousarray(dist5, np.float64).reshape(5)
    xyz = np.ascontiguousarray(xyz, np.float32).reshape(-1, 3)
    uv = np.zeros((len(xyz), 2), np.float32)
    rc = lib().orc_project_points(_p(Rt), _p(K4), _p(d5), _p(xyz), len(xyz), _p(uv))
    if rc: raise ValueError(rc)
    return uv


def frontend_size(w, h, target_width):
    dw, dh = C.c_int(0), C.c_int(0)
    rc = lib().orc_resize_scale_size(int(w), int(h), C.c_float(np.float32(target_width) / np.float32(w)),
                                     C.byref(dw), C.byref(dh))
    if rc: raise ValueError(rc)
    return dw.value, dh.value


def bgr2gray(bgr):
    bgr = np.ascontiguousarray(bgr, np.uint8)
    h, w, _ = bgr.shape
    out = np.zeros((h, w), np.uint8)
    rc = lib().orc_bgr2gray(_p(bgr), w, h, bgr.strides[0], _p(out), w)
    if rc: raise ValueError(rc)
    return out


def resize_linear(gray, scale):
    gray = _img(gray)
    h, w = gray.shape
    dw, dh = C.c_int(0), C.c_int(0)
    lib().orc_resize_scale_size(w, h, C.c_float(scale), C.byref(dw), C.byref(dh))
    out = np.zeros((dh.value, dw.value), np.uint8)
    rc = lib().orc_resize_linear(_p(gray), w, h, gray.strides[0], C.c_float(scale), _p(out), dw.value, dh.value, dw.value)
    if rc: raise ValueError(rc)
    return out


def linear_taps(src_size, dst_size, inv_scale, clamp_frac):
    ofs = np.zeros(dst_size, np.int32); taps = np.zeros((dst_size, 2), np.int16)
    lib().orc_linear_taps.argtypes = [C.c_int, C.c_int, C.c_double, C.c_int, C.c_void_p, C.c_void_p]
    rc = lib().orc_linear_taps(src_size, dst_size, float(inv_scale), int(clamp_frac), _p(ofs), _p(taps))
    if rc: raise ValueError(rc)
    return ofs, taps


def frontend(bgr, target_width):
    """main.cpp:124-129: BGR frame -> grey -> resized to target_width (aspect kept)."""
    bgr = np.ascontiguousarray(bgr, np.uint8)
    h, w, _ = bgr.shape
    dw, dh = frontend_size(w, h, target_width)
    out = np.zeros((dh, dw), np.uint8)
    rc = lib().orc_frontend(_p(bgr), w, h, bgr.strides[0], int(target_width), _p(out), dw, dh, dw)
    if rc: raise ValueError(rc)
    return out


def outlier_flags(xyz, matches, kp_q, kp_t, center_q, center_t, Rt_q, K4_q, dist5_q, Rt_t, K4_t, dist5_t):
    xyz = np.ascontiguousarray(xyz, np.float32).reshape(-1, 3)
    matches = np.ascontiguousarray(matches, DMATCH_DTYPE)
    kp_q = np.ascontiguousarray(kp_q, KEYPOINT_DTYPE); kp_t = np.ascontiguousarray(kp_t, KEYPOINT_DTYPE)
    cq = np.ascontiguousarray(center_q, np.float32).reshape(3); ct = np.ascontiguousarray(center_t, np.float32).reshape(3)
    f64 = lambda v, n: None if v is None else np.ascontiguousarray(v, np.float64).reshape(n)
    a = [f64(Rt_q, 12), f64(K4_q, 4), f64(dist5_q, 5), f64(Rt_t, 12), f64(K4_t, 4), f64(dist5_t, 5)]
    flags = np.zeros(len(xyz), np.uint8)
    fn = lib().orc_outlier_flags
    fn.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int] + [C.c_void_p] * 9
    rc = fn(_p(xyz), _p(matches), len(xyz), _p(kp_q), len(kp_q), _p(kp_t), len(kp_t), _p(cq), _p(ct),
            *[None if v is None else _p(v) for v in a], _p(flags))
    if rc: raise ValueError(rc)
    return flags


def duplicate_pairs(d1, p1, d2, p2, desc_dist=100, point_dist=1e-2):
    d1 = np.ascontiguousarray(d1, np.uint8).reshape(-1, 32)
    d2 = np.ascontiguousarray(d2, np.uint8).reshape(-1, 32)
    p1 = np.ascontiguousarray(p1, np.float32).reshape(-1, 3)
    p2 = np.ascontiguousarray(p2, np.float32).reshape(-1, 3)
    flag = np.zeros((len(d1), len(d2)), np.uint8)
    rc = lib().orc_duplicate_pairs(_p(d1), _p(p1), len(d1), _p(d2), _p(p2), len(d2),
                                   int(desc_dist), C.c_double(point_dist), _p(flag))
    if rc: raise ValueError(rc)
    return flag


def frame_scale_tables(scale_factor, nlevels):
    s = np.zeros(nlevels, np.float32); g = np.zeros(nlevels, np.float32); i = np.zeros(nlevels, np.float32)
    rc = lib().orc_frame_scale_tables(C.c_float(scale_factor), nlevels, _p(s), _p(g), _p(i))
    if rc: raise ValueError(rc)
    return s, g, i

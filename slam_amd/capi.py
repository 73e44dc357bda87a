"""ctypes binding of libslam_amd.so (the C ABI in include/slam_amd.h).

Used by the tests and the benchmark to drive the HIP library from Python.  There
is no fallback: a missing library raises at load time, a missing GPU makes the
create calls fail with SLAM_AMD_ERR_NO_DEVICE.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libslam_amd.so")

KEYPOINT_DTYPE = np.dtype(
    [("x", "<f4"), ("y", "<f4"), ("size", "<f4"), ("angle", "<f4"),
     ("response", "<f4"), ("octave", "<i4"), ("class_id", "<i4")]
)
DMATCH_DTYPE = np.dtype(
    [("queryIdx", "<i4"), ("trainIdx", "<i4"), ("imgIdx", "<i4"), ("distance", "<f4")]
)

OK, ERR_ARG, ERR_UNSUPPORTED, ERR_CAPACITY, ERR_NOMEM, ERR_HIP, ERR_NO_DEVICE = 0, -1, -2, -3, -4, -5, -6

# every symbol include/slam_amd.h declares
EXPORTS = [
    "slam_amd_version", "slam_amd_status_string", "slam_amd_last_error", "slam_amd_device_count",
    "slam_amd_default_orb_params",
    "slam_amd_extractor_create", "slam_amd_extractor_destroy", "slam_amd_extractor_levels", "slam_amd_extractor_set_grid",
    "slam_amd_extractor_set_quadtree",
    "slam_amd_extract", "slam_amd_extract_fetch", "slam_amd_extract_batch_async", "slam_amd_extractor_debug_level",
    "slam_amd_extractor_debug_blurred_level", "slam_amd_extractor_debug_set_select_cache",
    "slam_amd_matcher_create", "slam_amd_matcher_destroy", "slam_amd_matcher_reserve",
    "slam_amd_matcher_set_options", "slam_amd_matcher_debug_set_window_mode",
    "slam_amd_matcher_debug_set_sweep_kernel",
    "slam_amd_frame_match", "slam_amd_frame_match_async", "slam_amd_frame_match_noalloc_async",
    "slam_amd_frame_match_batch_async",
    "slam_amd_frame_match_chain_async",
    "slam_amd_match_pairs_async",
    "slam_amd_project_points", "slam_amd_project_points_async", "slam_amd_duplicate_pairs",
    "slam_amd_outlier_flags",
    "slam_amd_frontend_create", "slam_amd_frontend_destroy", "slam_amd_frontend_output_size",
    "slam_amd_frontend_run_async", "slam_amd_frontend_run", "slam_amd_linear_taps",
    "slam_amd_pipeline_create", "slam_amd_pipeline_destroy", "slam_amd_pipeline_input", "slam_amd_pipeline_submit",
    "slam_amd_pipeline_result", "slam_amd_pipeline_create_batched", "slam_amd_pipeline_flush",
    "slam_amd_prof_enable", "slam_amd_prof_collect", "slam_amd_prof_stage_name",
    "slam_amd_sincos_f32", "slam_amd_fast_atan2", "slam_amd_linear_exact_coeffs", "slam_amd_debug_pyramid_plan",
]
PROF_STAGES = 10


class OrbParams(C.Structure):
    _fields_ = [
        ("nfeatures", C.c_int32), ("scale_factor", C.c_float), ("nlevels", C.c_int32),
        ("edge_threshold", C.c_int32), ("first_level", C.c_int32), ("wta_k", C.c_int32),
        ("score_type", C.c_int32), ("patch_size", C.c_int32), ("fast_threshold", C.c_int32),
    ]


class SlamAmdError(RuntimeError):
    def __init__(self, status, message):
        super().__init__("slam_amd status %d: %s" % (status, message))
        self.status = status


_lib = None


def lib():
    """Load libslam_amd.so; raises OSError if it has not been built."""
    global _lib
    if _lib is None:
        path = os.environ.get("SLAM_AMD_LIB", LIB_PATH)   # override: experiments with other builds
        if not os.path.exists(path):
            raise OSError("%s not found: build it with `make -C slam_amd/csrc` "
                          "(or __graft_entry__.build())" % path)
        l = C.CDLL(path)
        l.slam_amd_status_string.restype = C.c_char_p
        l.slam_amd_last_error.restype = C.c_char_p
        l.slam_amd_fast_atan2.restype = C.c_float
        l.slam_amd_fast_atan2.argtypes = [C.c_float, C.c_float]
        l.slam_amd_sincos_f32.argtypes = [C.c_float, C.POINTER(C.c_float), C.POINTER(C.c_float)]
        l.slam_amd_sincos_f32.restype = None
        l.slam_amd_default_orb_params.restype = None
        l.slam_amd_matcher_reserve.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int]
        l.slam_amd_extract_fetch.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.POINTER(C.c_int)]
        l.slam_amd_matcher_set_options.argtypes = [C.c_void_p, C.c_float, C.c_int]
        l.slam_amd_matcher_debug_set_window_mode.argtypes = [C.c_void_p, C.c_int]
        l.slam_amd_matcher_debug_set_sweep_kernel.argtypes = [C.c_void_p, C.c_int]
        l.slam_amd_extract_batch_async.argtypes = [
            C.c_void_p, C.c_void_p, C.c_int, C.c_size_t, C.c_int, C.c_void_p, C.c_void_p,
            C.c_void_p, C.c_int, C.c_void_p]
        l.slam_amd_frame_match_async.argtypes = [
            C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int,
            C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_float, C.c_float, C.c_int,
            C.c_void_p, C.c_void_p, C.c_void_p]
        l.slam_amd_frame_match_noalloc_async.argtypes = l.slam_amd_frame_match_async.argtypes
        l.slam_amd_frame_match_batch_async.argtypes = [
            C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
            C.c_int, C.c_void_p, C.c_float, C.c_float, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        l.slam_amd_frame_match_chain_async.argtypes = [
            C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p,
            C.c_float, C.c_float, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        l.slam_amd_match_pairs_async.argtypes = [
            C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int,
            C.c_float, C.c_float, C.c_int, C.c_void_p, C.c_void_p]
        l.slam_amd_frame_match.argtypes = [
            C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int,
            C.c_void_p, C.c_int, C.c_void_p, C.c_float, C.c_float, C.c_int, C.c_void_p,
            C.POINTER(C.c_int)]
        l.slam_amd_extract.argtypes = [
            C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int,
            C.POINTER(C.c_int)]
        l.slam_amd_extractor_create.argtypes = [
            C.POINTER(OrbParams), C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_void_p)]
        l.slam_amd_extractor_destroy.argtypes = [C.c_void_p]
        l.slam_amd_extractor_levels.argtypes = [C.c_void_p] + [C.c_void_p] * 5
        l.slam_amd_extractor_set_grid.argtypes = [C.c_void_p, C.c_int]
        l.slam_amd_extractor_set_quadtree.argtypes = [C.c_void_p, C.c_int]
        l.slam_amd_extractor_debug_level.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int]
        l.slam_amd_extractor_debug_blurred_level.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int]
        l.slam_amd_extractor_debug_set_select_cache.argtypes = [C.c_void_p, C.c_int]
        l.slam_amd_matcher_create.argtypes = [C.c_int, C.POINTER(C.c_void_p)]
        l.slam_amd_matcher_destroy.argtypes = [C.c_void_p]
        l.slam_amd_project_points.argtypes = [C.c_void_p] * 5 + [C.c_int, C.c_void_p]
        l.slam_amd_project_points_async.argtypes = [C.c_void_p] * 5 + [C.c_int, C.c_void_p, C.c_void_p]
        l.slam_amd_duplicate_pairs.argtypes = [
            C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int,
            C.c_int, C.c_double, C.c_void_p]
        l.slam_amd_linear_exact_coeffs.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        l.slam_amd_pipeline_create_batched.argtypes = [
            C.POINTER(OrbParams), C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float, C.c_int,
            C.POINTER(C.c_void_p)]
        l.slam_amd_pipeline_destroy.argtypes = [C.c_void_p]
        l.slam_amd_pipeline_input.argtypes = [C.c_void_p, C.c_longlong, C.POINTER(C.c_void_p), C.POINTER(C.c_int)]
        l.slam_amd_pipeline_submit.argtypes = [C.c_void_p, C.c_longlong]
        l.slam_amd_pipeline_flush.argtypes = [C.c_void_p]
        l.slam_amd_pipeline_result.argtypes = [C.c_void_p, C.c_longlong, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p),
                                               C.POINTER(C.c_int), C.POINTER(C.c_void_p), C.POINTER(C.c_int)]
        l.slam_amd_prof_stage_name.restype = C.c_char_p
        l.slam_amd_prof_collect.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
        _lib = l
    return _lib


DIST_LIB_PATH = os.path.join(_HERE, "libslam_amd_dist.so")
DIST_EXPORTS = [
    "slam_amd_pair_shard_range", "slam_amd_pair_shard_chunk", "slam_amd_pair_shard_unique_id",
    "slam_amd_pair_shard_create", "slam_amd_pair_shard_destroy", "slam_amd_pair_shard_run",
    "slam_amd_pair_shard_wait", "slam_amd_pair_shard_result",
]
UNIQUE_ID_BYTES = 128
_dist = None


def dist_lib():
    """Load libslam_amd_dist.so (the pair shard over RCCL); the core library first."""
    global _dist
    if _dist is None:
        lib()
        if not os.path.exists(DIST_LIB_PATH):
            raise OSError("%s not found: build it with `make -C slam_amd/csrc`" % DIST_LIB_PATH)
        d = C.CDLL(DIST_LIB_PATH)
        d.slam_amd_pair_shard_create.argtypes = [C.c_void_p, C.c_int, C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                                 C.POINTER(C.c_void_p)]
        d.slam_amd_pair_shard_destroy.argtypes = [C.c_void_p]
        d.slam_amd_pair_shard_run.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int,
                                              C.c_float, C.c_float, C.c_int, C.c_void_p]
        d.slam_amd_pair_shard_wait.argtypes = [C.c_void_p, C.c_void_p]
        d.slam_amd_pair_shard_result.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_int)]
        d.slam_amd_pair_shard_unique_id.argtypes = [C.c_char_p]
        _dist = d
    return _dist


def pair_shard_range(n_pairs, world, rank):
    lo, hi, per = C.c_int(0), C.c_int(0), C.c_int(0)
    check(dist_lib().slam_amd_pair_shard_range(n_pairs, world, rank, C.byref(lo), C.byref(hi), C.byref(per)))
    return lo.value, hi.value, per.value


def pair_shard_chunk(per, chunks, chunk):
    a, b = C.c_int(0), C.c_int(0)
    check(dist_lib().slam_amd_pair_shard_chunk(per, chunks, chunk, C.byref(a), C.byref(b)))
    return a.value, b.value


def pair_shard_unique_id():
    buf = C.create_string_buffer(UNIQUE_ID_BYTES)
    check(dist_lib().slam_amd_pair_shard_unique_id(buf))
    return buf.raw


class PairShard:
    """Owner of a slam_amd_pair_shard handle (keyframe-pair matching of one rank + gather to rank 0 over RCCL)."""

    def __init__(self, matcher, device, unique_id, world, rank, n_pairs, n_per, chunks):
        self._h = C.c_void_p()
        self.world, self.rank, self.n_pairs, self.n_per = world, rank, n_pairs, n_per
        self.lo, self.hi, self.per = pair_shard_range(n_pairs, world, rank)
        check(dist_lib().slam_amd_pair_shard_create(matcher._h, device, unique_id, world, rank, n_pairs, n_per, chunks,
                                                    C.byref(self._h)))

    def close(self):
        if self._h:
            dist_lib().slam_amd_pair_shard_destroy(self._h)
            self._h = C.c_void_p()

    def run(self, d_desc, d_kp, n_keyframes, d_pairs_local, max_dist=300.0, area=-1.0, max_level=-1, stream=0, limit_rows=-1):
        check(dist_lib().slam_amd_pair_shard_run(self._h, _ptr(d_desc), _ptr(d_kp), n_keyframes, _ptr(d_pairs_local),
                                                 int(limit_rows), max_dist, area, int(max_level), C.c_void_p(stream)))

    def wait(self, stream=0):
        check(dist_lib().slam_amd_pair_shard_wait(self._h, C.c_void_p(stream)))

    def result_ptr(self):
        """Rank 0: device address of the [n_pairs][n_per] uint32 table; other ranks: None."""
        p = C.c_void_p()
        n = C.c_int(0)
        check(dist_lib().slam_amd_pair_shard_result(self._h, C.byref(p), C.byref(n)))
        return p.value


def check(status):
    if status != OK:
        raise SlamAmdError(status, lib().slam_amd_last_error().decode())


def orb_params(nfeatures=1000, scale_factor=1.2, nlevels=8, edge_threshold=31, fast_threshold=20,
               first_level=0, wta_k=2, score_type=1, patch_size=31):
    return OrbParams(nfeatures, scale_factor, nlevels, edge_threshold, first_level, wta_k,
                     score_type, patch_size, fast_threshold)


def _ptr(a):
    if a is None:
        return None
    if isinstance(a, np.ndarray):
        return a.ctypes.data_as(C.c_void_p)
    if isinstance(a, int):
        return C.c_void_p(a)
    if hasattr(a, "data_ptr"):  # torch tensor
        return C.c_void_p(a.data_ptr())
    raise TypeError(type(a))


def linear_taps(src_size, dst_size, inv_scale, clamp_frac):
    ofs = np.zeros(dst_size, np.int32); taps = np.zeros((dst_size, 2), np.int16)
    lib().slam_amd_linear_taps.argtypes = [C.c_int, C.c_int, C.c_double, C.c_int, C.c_void_p, C.c_void_p]
    check(lib().slam_amd_linear_taps(src_size, dst_size, float(inv_scale), int(clamp_frac), _ptr(ofs), _ptr(taps)))
    return ofs, taps


def pyramid_plan(params, width, height):
    """Launch plan of the pyramid kernel (host computation): list of dicts per launch with
    first, last, tile, ntx, nty, lds_bytes, spanx [ntx][slots][4], spany [nty][slots][4]."""
    n = C.c_int(0)
    fn = lib().slam_amd_debug_pyramid_plan
    fn.argtypes = [C.POINTER(OrbParams), C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
    check(fn(C.byref(params), width, height, -1, C.byref(n), None, None, None, 0))
    out = []
    for g in range(n.value):
        info = np.zeros(6, np.int32)
        check(fn(C.byref(params), width, height, g, None, _ptr(info), None, None, 0))
        first, last, tile, ntx, nty, lds = [int(v) for v in info]
        slots = last - first + 2
        sx = np.zeros((ntx, slots, 4), np.int16); sy = np.zeros((nty, slots, 4), np.int16)
        check(fn(C.byref(params), width, height, g, None, None, _ptr(sx), _ptr(sy), max(sx.size, sy.size)))
        out.append(dict(first=first, last=last, tile=tile, ntx=ntx, nty=nty, lds_bytes=lds, spanx=sx, spany=sy))
    return out


def sincos_f32(angle_rad):
    c = C.c_float(); s = C.c_float()
    lib().slam_amd_sincos_f32(C.c_float(angle_rad), C.byref(c), C.byref(s))
    return c.value, s.value


def fast_atan2(y, x):
    return float(lib().slam_amd_fast_atan2(float(y), float(x)))


def linear_exact_coeffs(src, dst):
    off = np.zeros(dst, np.int32); c1 = np.zeros(dst, np.uint16)
    check(lib().slam_amd_linear_exact_coeffs(src, dst, _ptr(off), _ptr(c1)))
    return off, c1


def prof_enable(max_marks):
    check(lib().slam_amd_prof_enable(int(max_marks)))


def prof_collect():
    """{stage name: (total ms, calls)} of the stages recorded since the last collect."""
    ms = np.zeros(PROF_STAGES, np.float64); calls = np.zeros(PROF_STAGES, np.int32)
    check(lib().slam_amd_prof_collect(_ptr(ms), _ptr(calls), PROF_STAGES))
    return {lib().slam_amd_prof_stage_name(i).decode(): (float(ms[i]), int(calls[i]))
            for i in range(PROF_STAGES) if calls[i]}


def device_count():
    n = C.c_int(0)
    check(lib().slam_amd_device_count(C.byref(n)))
    return n.value


class Pipeline:
    """Owner of a slam_amd_pipeline handle: the C++ host of the streaming configuration (frames in pinned
    host memory; upload, extract + match against the previous frame and download on three HIP streams over a
    ring of `depth` slots of `batch` frames each)."""

    def __init__(self, params, width, height, capacity, depth=3, batch=1, device=0, max_dist=300.0, area=-1.0, max_level=-1):
        self._h = C.c_void_p()
        self.width, self.height, self.capacity, self.depth, self.batch = width, height, capacity, depth, batch
        check(lib().slam_amd_pipeline_create_batched(C.byref(params), width, height, capacity, depth, batch, device,
                                                     max_dist, area, int(max_level), C.byref(self._h)))

    def close(self):
        if self._h:
            lib().slam_amd_pipeline_destroy(self._h)
            self._h = C.c_void_p()

    def input(self, frame):
        """The pinned host buffer of frame `frame` (the next one to submit) as a (height, width) uint8 array;
        waits until the slot is free."""
        buf, stride = C.c_void_p(), C.c_int(0)
        check(lib().slam_amd_pipeline_input(self._h, frame, C.byref(buf), C.byref(stride)))
        raw = (C.c_uint8 * (self.height * stride.value)).from_address(buf.value)
        return np.frombuffer(raw, dtype=np.uint8).reshape(self.height, stride.value)[:, :self.width]

    def submit(self, frame):
        check(lib().slam_amd_pipeline_submit(self._h, frame))

    def flush(self):
        check(lib().slam_amd_pipeline_flush(self._h))

    def result(self, frame, copy=True):
        """Waits for frame `frame`: (keypoints, descriptors [n][32], matches against the frame before)."""
        kp, desc, mt = C.c_void_p(), C.c_void_p(), C.c_void_p()
        n, nm = C.c_int(0), C.c_int(0)
        check(lib().slam_amd_pipeline_result(self._h, frame, C.byref(kp), C.byref(desc), C.byref(n), C.byref(mt), C.byref(nm)))

        def view(addr, count, dtype):
            if count == 0:
                return np.zeros(0, dtype)
            raw = (C.c_uint8 * (count * np.dtype(dtype).itemsize)).from_address(addr)
            a = np.frombuffer(raw, dtype=dtype)
            return a.copy() if copy else a
        return (view(kp.value, n.value, KEYPOINT_DTYPE), view(desc.value, n.value * 32, np.uint8).reshape(-1, 32),
                view(mt.value, nm.value, DMATCH_DTYPE))


class Extractor:
    """Owner of a slam_amd_extractor handle."""

    def __init__(self, params, width, height, max_batch=1, max_keypoints=None, device=0):
        self.params = params
        self.width, self.height = width, height
        self.max_batch = max_batch
        self.max_keypoints = max_keypoints or max(4 * params.nfeatures, 1024)
        self.device = device
        self._h = C.c_void_p()
        check(lib().slam_amd_extractor_create(C.byref(params), width, height, max_batch,
                                               self.max_keypoints, device, C.byref(self._h)))

    def close(self):
        if self._h:
            lib().slam_amd_extractor_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_grid(self, cell):
        """Grid distribution of keypoints (not in the reference): cell 8, 16 or 32; 0 = off."""
        check(lib().slam_amd_extractor_set_grid(self._h, cell))

    def set_quadtree(self, on):
        """Quadtree distribution with an exact per-level count (not in the reference); False = off."""
        check(lib().slam_amd_extractor_set_quadtree(self._h, int(bool(on))))

    def levels(self):
        n = C.c_int(0)
        ws = np.zeros(16, np.int32); hs = np.zeros(16, np.int32); qs = np.zeros(16, np.int32)
        sc = np.zeros(16, np.float32)
        check(lib().slam_amd_extractor_levels(self._h, C.cast(C.byref(n), C.c_void_p), _ptr(ws),
                                               _ptr(hs), _ptr(qs), _ptr(sc)))
        k = n.value
        return ws[:k].copy(), hs[:k].copy(), qs[:k].copy(), sc[:k].copy()

    def extract(self, img, capacity=None):
        """Host, blocking: numpy image in, (keypoints, descriptors) out."""
        img = np.ascontiguousarray(img, np.uint8)
        h, w = img.shape
        cap = capacity if capacity is not None else self.max_keypoints
        kps = np.zeros(max(cap, 1), KEYPOINT_DTYPE)
        desc = np.zeros((max(cap, 1), 32), np.uint8)
        n = C.c_int(0)
        check(lib().slam_amd_extract(self._h, _ptr(img), w, h, img.strides[0], _ptr(kps), _ptr(desc),
                                      cap, C.byref(n)))
        return kps[:n.value].copy(), desc[:n.value].copy()

    def extract_fetch(self, capacity):
        """Results of the last extract() call again (e.g. after it ended in ERR_CAPACITY)."""
        kps = np.zeros(max(capacity, 1), KEYPOINT_DTYPE)
        desc = np.zeros((max(capacity, 1), 32), np.uint8)
        n = C.c_int(0)
        check(lib().slam_amd_extract_fetch(self._h, _ptr(kps), _ptr(desc), capacity, C.byref(n)))
        return kps[:n.value].copy(), desc[:n.value].copy()

    def extract_batch_async(self, d_images, batch, frame_pitch, row_stride, d_kp, d_desc, d_count,
                            capacity, stream=0):
        check(lib().slam_amd_extract_batch_async(self._h, _ptr(d_images), batch, frame_pitch, row_stride,
                                                  _ptr(d_kp), _ptr(d_desc), _ptr(d_count), capacity,
                                                  C.c_void_p(stream)))

    def debug_set_select_cache(self, entries):
        check(lib().slam_amd_extractor_debug_set_select_cache(self._h, entries))

    def debug_level(self, frame, level, blurred=False):
        ws, hs, _, _ = self.levels()
        out = np.zeros((hs[level], ws[level]), np.uint8)
        fn = lib().slam_amd_extractor_debug_blurred_level if blurred else lib().slam_amd_extractor_debug_level
        check(fn(self._h, frame, level, _ptr(out), int(ws[level])))
        return out


class Frontend:
    """BGR (or grey) frame -> grey frame of the working width: cvtColor + resize of main.cpp:124-129."""

    def __init__(self, src_width, src_height, target_width, channels=3, device=0):
        self._h = C.c_void_p()
        l = lib()
        l.slam_amd_frontend_create.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]
        l.slam_amd_frontend_destroy.argtypes = [C.c_void_p]
        l.slam_amd_frontend_output_size.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        l.slam_amd_frontend_run.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int]
        l.slam_amd_frontend_run_async.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_size_t, C.c_int,
                                                  C.c_void_p, C.c_size_t, C.c_int, C.c_void_p]
        check(l.slam_amd_frontend_create(src_width, src_height, channels, target_width, device, C.byref(self._h)))
        self.src_width, self.src_height, self.channels = src_width, src_height, channels
        w, h = C.c_int(0), C.c_int(0)
        check(l.slam_amd_frontend_output_size(self._h, C.byref(w), C.byref(h)))
        self.width, self.height = w.value, h.value

    def close(self):
        if self._h:
            lib().slam_amd_frontend_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def run(self, frame):
        """Host, blocking: numpy [h, w, 3] (or [h, w]) uint8 in, grey [height, width] out."""
        frame = np.ascontiguousarray(frame, np.uint8)
        assert frame.shape[0] == self.src_height and frame.shape[1] == self.src_width
        out = np.zeros((self.height, self.width), np.uint8)
        check(lib().slam_amd_frontend_run(self._h, _ptr(frame), frame.strides[0], _ptr(out), self.width))
        return out

    def run_async(self, d_src, batch, src_frame_pitch, src_row_stride, d_gray, dst_frame_pitch, dst_row_stride,
                  stream=0):
        check(lib().slam_amd_frontend_run_async(self._h, _ptr(d_src), batch, src_frame_pitch, src_row_stride,
                                                _ptr(d_gray), dst_frame_pitch, dst_row_stride, _ptr(stream)))


class Matcher:
    """Owner of a slam_amd_matcher handle."""

    def __init__(self, device=0):
        self._h = C.c_void_p()
        check(lib().slam_amd_matcher_create(device, C.byref(self._h)))

    def close(self):
        if self._h:
            lib().slam_amd_matcher_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def reserve(self, batch, query_rows, train_rows=0):
        """Size the device scratch for calls of at most `batch` problems of query_rows x train_rows rows
        (train_rows only matters for the cell-sorted windowed path).  Do this before capturing a HIP graph."""
        check(lib().slam_amd_matcher_reserve(self._h, int(batch), int(query_rows), int(train_rows)))

    def debug_set_window_mode(self, mode):
        check(lib().slam_amd_matcher_debug_set_window_mode(self._h, mode))

    def debug_set_sweep_kernel(self, mode):
        check(lib().slam_amd_matcher_debug_set_sweep_kernel(self._h, mode))

    def set_options(self, ratio=0.0, histo_length=0):
        """Ratio test and rotation-consistency filter (neither is in the reference; 0 = off)."""
        check(lib().slam_amd_matcher_set_options(self._h, ratio, histo_length))

    def frame_match(self, d1, kp1, d2, kp2, ids=None, q_xy=None, max_dist=300.0, area=-1.0, max_level=4):
        """Host, blocking (numpy in, numpy DMatch records out)."""
        d1 = np.ascontiguousarray(d1, np.uint8).reshape(-1, 32)
        d2 = np.ascontiguousarray(d2, np.uint8).reshape(-1, 32)
        kp1 = None if kp1 is None else np.ascontiguousarray(kp1, KEYPOINT_DTYPE)
        kp2 = None if kp2 is None else np.ascontiguousarray(kp2, KEYPOINT_DTYPE)
        ids_a = None if ids is None or len(ids) == 0 else np.ascontiguousarray(ids, np.int32)
        xy = None if q_xy is None else np.ascontiguousarray(q_xy, np.float32)
        nq = len(d1) if ids_a is None else len(ids_a)
        out = np.zeros(max(nq, 1), DMATCH_DTYPE)
        n = C.c_int(0)
        check(lib().slam_amd_frame_match(self._h, _ptr(d1), _ptr(kp1), len(d1), _ptr(d2), _ptr(kp2), len(d2),
                                          _ptr(ids_a), 0 if ids_a is None else len(ids_a), _ptr(xy),
                                          max_dist, area, int(max_level), _ptr(out), C.byref(n)))
        return out[:n.value].copy()

    def frame_match_async(self, d_desc1, d_kp1, n1, d_n1, d_desc2, d_kp2, n2, d_n2, d_out, d_n_out,
                          d_ids=None, n_ids=0, d_q_xy=None, max_dist=300.0, area=-1.0, max_level=4,
                          stream=0):
        check(lib().slam_amd_frame_match_async(self._h, _ptr(d_desc1), _ptr(d_kp1), n1, _ptr(d_n1),
                                                _ptr(d_desc2), _ptr(d_kp2), n2, _ptr(d_n2),
                                                _ptr(d_ids), n_ids, _ptr(d_q_xy), max_dist, area,
                                                int(max_level), _ptr(d_out), _ptr(d_n_out),
                                                C.c_void_p(stream)))

    def frame_match_noalloc_async(self, d_desc1, d_kp1, n1, d_n1, d_desc2, d_kp2, n2, d_n2, d_out, d_n_out,
                                  d_ids=None, n_ids=0, d_q_xy=None, max_dist=300.0, area=-1.0, max_level=4,
                                  stream=0):
        """frame_match_async on the library's static device scratch: allocates nothing."""
        check(lib().slam_amd_frame_match_noalloc_async(self._h, _ptr(d_desc1), _ptr(d_kp1), n1, _ptr(d_n1),
                                                        _ptr(d_desc2), _ptr(d_kp2), n2, _ptr(d_n2),
                                                        _ptr(d_ids), n_ids, _ptr(d_q_xy), max_dist, area,
                                                        int(max_level), _ptr(d_out), _ptr(d_n_out),
                                                        C.c_void_p(stream)))

    def frame_match_batch_async(self, batch, d_desc1, d_kp1, cap1, d_n1, d_desc2, d_kp2, cap2, d_n2,
                                d_out, d_n_out, max_dist=300.0, area=-1.0, max_level=4, stream=0):
        check(lib().slam_amd_frame_match_batch_async(self._h, batch, _ptr(d_desc1), _ptr(d_kp1), cap1,
                                                      _ptr(d_n1), _ptr(d_desc2), _ptr(d_kp2), cap2,
                                                      _ptr(d_n2), max_dist, area, int(max_level),
                                                      _ptr(d_out), _ptr(d_n_out), C.c_void_p(stream)))

    def frame_match_chain_async(self, batch, d_prev_desc, d_prev_kp, d_prev_n, d_desc, d_kp, cap, d_n,
                                d_out, d_n_out, max_dist=300.0, area=-1.0, max_level=4, stream=0):
        """Frame b of the batch against the frame before it (frame 0 against the one at d_prev_*)."""
        check(lib().slam_amd_frame_match_chain_async(self._h, batch, _ptr(d_prev_desc), _ptr(d_prev_kp),
                                                      _ptr(d_prev_n), _ptr(d_desc), _ptr(d_kp), cap, _ptr(d_n),
                                                      max_dist, area, int(max_level), _ptr(d_out), _ptr(d_n_out),
                                                      C.c_void_p(stream)))

    def match_pairs_async(self, d_desc, d_kp, n_keyframes, n_per, d_pairs, n_pairs, d_out,
                          max_dist=300.0, area=-1.0, max_level=-1, stream=0):
        check(lib().slam_amd_match_pairs_async(self._h, _ptr(d_desc), _ptr(d_kp), n_keyframes, n_per,
                                                _ptr(d_pairs), n_pairs, max_dist, area, int(max_level),
                                                _ptr(d_out), C.c_void_p(stream)))

    def project_points(self, Rt, K4, dist5, xyz):
        Rt = np.ascontiguousarray(Rt, np.float64).reshape(12)
        K4 = np.ascontiguousarray(K4, np.float64).reshape(4)
        d5 = None if dist5 is None else np.ascontiguousarray(dist5, np.float64).reshape(5)
        xyz = np.ascontiguousarray(xyz, np.float32).reshape(-1, 3)
        uv = np.zeros((len(xyz), 2), np.float32)
        check(lib().slam_amd_project_points(self._h, _ptr(Rt), _ptr(K4), _ptr(d5), _ptr(xyz), len(xyz),
                                             _ptr(uv)))
        return uv

    def outlier_flags(self, xyz, matches, kp_q, kp_t, center_q, center_t, Rt_q, K4_q, dist5_q, Rt_t, K4_t, dist5_t):
        """isOutlier of the reference for every (point, match): uint8 flags, 1 = outlier."""
        xyz = np.ascontiguousarray(xyz, np.float32).reshape(-1, 3)
        matches = np.ascontiguousarray(matches, DMATCH_DTYPE)
        kp_q = np.ascontiguousarray(kp_q, KEYPOINT_DTYPE); kp_t = np.ascontiguousarray(kp_t, KEYPOINT_DTYPE)
        f32 = lambda v: np.ascontiguousarray(v, np.float32).reshape(3)
        f64 = lambda v, n: None if v is None else np.ascontiguousarray(v, np.float64).reshape(n)
        cq, ct = f32(center_q), f32(center_t)
        a = [f64(Rt_q, 12), f64(K4_q, 4), f64(dist5_q, 5), f64(Rt_t, 12), f64(K4_t, 4), f64(dist5_t, 5)]
        flags = np.zeros(len(xyz), np.uint8)
        fn = lib().slam_amd_outlier_flags
        fn.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int] + \
                      [C.c_void_p] * 9
        check(fn(self._h, _ptr(xyz), _ptr(matches), len(xyz), _ptr(kp_q), len(kp_q), _ptr(kp_t), len(kp_t),
                 _ptr(cq), _ptr(ct), *[None if v is None else _ptr(v) for v in a], _ptr(flags)))
        return flags

    def duplicate_pairs(self, d1, p1, d2, p2, desc_dist=100, point_dist=1e-2):
        d1 = np.ascontiguousarray(d1, np.uint8).reshape(-1, 32)
        d2 = np.ascontiguousarray(d2, np.uint8).reshape(-1, 32)
        p1 = np.ascontiguousarray(p1, np.float32).reshape(-1, 3)
        p2 = np.ascontiguousarray(p2, np.float32).reshape(-1, 3)
        flag = np.zeros((len(d1), len(d2)), np.uint8)
        check(lib().slam_amd_duplicate_pairs(self._h, _ptr(d1), _ptr(p1), len(d1), _ptr(d2), _ptr(p2),
                                              len(d2), int(desc_dist), float(point_dist), _ptr(flag)))
        return flag

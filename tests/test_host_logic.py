"""CPU-side tests of the library's host logic and of its C ABI surface (no GPU)."""
import ctypes
import math
import os
import re

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def test_library_exports_every_declared_symbol(capi):
    hdr = open(os.path.join(ROOT, "include", "slam_amd.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(slam_amd_[a-z0-9_]+)\s*\(", hdr))
    assert len(declared) >= 25
    lib = capi.lib()
    missing = [s for s in sorted(declared) if not hasattr(lib, s)]
    assert not missing, missing
    assert declared == set(capi.EXPORTS), declared ^ set(capi.EXPORTS)


def test_record_layouts(capi):
    assert capi.KEYPOINT_DTYPE.itemsize == 28      # cv::KeyPoint
    assert capi.DMATCH_DTYPE.itemsize == 16        # cv::DMatch
    assert ctypes.sizeof(capi.OrbParams) == 36


def test_status_strings_and_defaults(capi):
    lib = capi.lib()
    assert lib.slam_amd_version() == 100
    assert lib.slam_amd_status_string(0) == b"ok"
    assert b"capacity" in lib.slam_amd_status_string(capi.ERR_CAPACITY)
    p = capi.OrbParams()
    lib.slam_amd_default_orb_params(ctypes.byref(p))
    # main.cpp:100: cv::ORB::create(1000, 1.2f, 8, 31, 0, 2, cv::ORB::FAST_SCORE)
    assert (p.nfeatures, p.nlevels, p.edge_threshold, p.first_level, p.wta_k, p.score_type,
            p.patch_size, p.fast_threshold) == (1000, 8, 31, 0, 2, 1, 31, 20)
    assert p.scale_factor == np.float32(1.2)


def test_no_gpu_means_loud_failure(capi):
    if capi.device_count() > 0:
        pytest.skip("a GPU is visible")
    with pytest.raises(capi.SlamAmdError) as e:
        capi.Extractor(capi.orb_params(), 640, 480)
    assert e.value.status == capi.ERR_NO_DEVICE
    with pytest.raises(capi.SlamAmdError) as e:
        capi.Matcher()
    assert e.value.status == capi.ERR_NO_DEVICE


def test_unsupported_variants_are_rejected_before_touching_the_gpu(capi):
    for kw in (dict(wta_k=3), dict(score_type=0), dict(patch_size=25), dict(first_level=1),
               dict(edge_threshold=19)):
        with pytest.raises(capi.SlamAmdError) as e:
            capi.Extractor(capi.orb_params(**kw), 640, 480)
        assert e.value.status == capi.ERR_UNSUPPORTED, kw
    for kw in (dict(nlevels=0), dict(nlevels=17), dict(scale_factor=1.0), dict(fast_threshold=300)):
        with pytest.raises(capi.SlamAmdError) as e:
            capi.Extractor(capi.orb_params(**kw), 640, 480)
        assert e.value.status == capi.ERR_ARG, kw


def test_sincos_matches_libm_on_every_float_angle():
    """The descriptor kernel's cos / sin (double polynomial, rounded to float) against libm's, for
    every float from 0 to just above 2 pi (1.09e9 values; tests/cpp/sweep_sincos.cpp, a few seconds
    on 8 cores): one differing last bit could flip a rounded pattern coordinate."""
    import subprocess
    cpp = os.path.join(os.path.dirname(os.path.abspath(__file__)), "cpp")
    subprocess.check_call(["make", "-C", cpp, "sweep_sincos"], stdout=subprocess.DEVNULL)
    r = subprocess.run([os.path.join(cpp, "sweep_sincos")], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and r.stdout.startswith("0 of 108"), r.stdout + r.stderr


def test_fast_atan2_matches_oracle_bitwise(capi, oracle):
    rng = np.random.default_rng(1)
    ys = rng.integers(-1400000, 1400000, 200000)
    xs = rng.integers(-1400000, 1400000, 200000)
    ys[:5] = [0, 0, 5, -5, 0]; xs[:5] = [0, 5, 0, 0, -5]
    for y, x in zip(ys.tolist(), xs.tolist()):
        assert capi.fast_atan2(y, x) == oracle.fast_atan2(y, x)


def test_resize_tables_match_oracle(capi, oracle):
    for src, dst in [(640, 533), (533, 444), (444, 370), (480, 400), (1241, 1034), (3840, 3200),
                     (2160, 1800), (179, 149), (100, 100), (7, 3), (3, 7), (2, 1), (1, 1)]:
        a_off, a_c = capi.linear_exact_coeffs(src, dst)
        b_off, b_c = oracle.linear_exact_coeffs(src, dst)
        assert np.array_equal(a_off, b_off) and np.array_equal(a_c, b_c), (src, dst)


def test_pair_sharding_covers_every_pair_once():
    from slam_amd import pairs
    p = pairs.all_pairs(9)
    assert len(p) == 36 and (p[:, 0] < p[:, 1]).all() and len({tuple(r) for r in p.tolist()}) == 36
    for n_pairs in (0, 1, 7, 36, 523776):
        for world in (1, 2, 3, 4, 8):
            seen = 0
            per0 = None
            for r in range(world):
                lo, hi, per = pairs.shard(n_pairs, world, r)
                per0 = per if per0 is None else per0
                assert per == per0 and 0 <= hi - lo <= per and lo == min(r * per, n_pairs)
                seen += hi - lo
            assert seen == n_pairs
    for per in (0, 1, 5, 64, 65473):
        for chunks in (1, 3, 8):
            edges = [pairs.chunk_rows(per, chunks, c) for c in range(chunks)]
            assert edges[0][0] == 0 and edges[-1][1] == per
            assert all(edges[i][1] == edges[i + 1][0] for i in range(chunks - 1))


def test_frontend_resize_tables_match_oracle(oracle):
    """The library builds the cv::resize(INTER_LINEAR) tables on the host; same bits as the oracle."""
    from slam_amd import capi
    for src, dst_w in ((1920, 1080), (1280, 1080), (640, 1080), (123, 77), (3840, 640)):
        sc = np.float32(dst_w) / np.float32(src)
        dst = int(np.rint(src * np.float64(sc)))
        for clamp in (0, 1):
            o1, t1 = capi.linear_taps(src, dst, float(sc), clamp)
            o2, t2 = oracle.linear_taps(src, dst, float(sc), clamp)
            assert np.array_equal(o1, o2) and np.array_equal(t1, t2), (src, dst_w, clamp)


def test_frontend_arguments_are_checked_before_touching_the_gpu(capi):
    for args, status in (((0, 480, 320, 3), capi.ERR_ARG), ((640, 480, 0, 3), capi.ERR_ARG),
                         ((40000, 480, 320, 3), capi.ERR_ARG), ((640, 480, 320, 4), capi.ERR_UNSUPPORTED),
                         ((640, 480, 320, 2), capi.ERR_UNSUPPORTED)):
        with pytest.raises(capi.SlamAmdError) as e:
            capi.Frontend(args[0], args[1], args[2], channels=args[3])
        assert e.value.status == status, args
    if capi.device_count() == 0:
        with pytest.raises(capi.SlamAmdError) as e:
            capi.Frontend(1920, 1080, 1080)
        assert e.value.status == capi.ERR_NO_DEVICE
    with pytest.raises(capi.SlamAmdError):
        capi.linear_taps(0, 10, 0.5, 1)
    with pytest.raises(capi.SlamAmdError):
        capi.linear_taps(10, 10, -1.0, 1)


def test_dist_library_exports_and_share_arithmetic():
    """libslam_amd_dist.so loads, exports what include/slam_amd_dist.h declares, and cuts the pair
    list exactly as the Python driver of the same exchange (slam_amd/pairs.py) does."""
    from slam_amd import capi, pairs
    d = capi.dist_lib()
    assert not [s for s in capi.DIST_EXPORTS if not hasattr(d, s)]
    for n_pairs in (0, 1, 15, 523776):
        for world in (1, 2, 3, 8):
            for rank in range(world):
                assert capi.pair_shard_range(n_pairs, world, rank) == pairs.shard(n_pairs, world, rank)
            per = pairs.shard(n_pairs, world, 0)[2]
            for chunks in (1, 4, 16):
                for c in range(chunks):
                    assert capi.pair_shard_chunk(per, chunks, c) == pairs.chunk_rows(per, chunks, c)

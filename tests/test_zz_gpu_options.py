"""GPU parity of the newest entry points (grid distribution, batched isOutlier).

Kept in a file that pytest collects after the others, so that under `pytest -x` a
failure here cannot hide the tests of the reference path."""
import numpy as np
import pytest

from slam_amd import synth

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("cell", [8, 16, 32])
def test_grid_distribution_option_vs_oracle(oracle, capi, cell):
    """Keypoint grid distribution (an option the reference does not have) against its oracle."""
    for (w, h, nf, seed) in ((640, 480, 2000, 1), (333, 271, 500, 9)):
        img = synth.scene(w, h, seed=seed)
        po = oracle.params(nfeatures=nf)
        ex = capi.Extractor(capi.orb_params(nfeatures=nf), w, h, max_batch=1, max_keypoints=8192)
        ex.set_grid(cell)
        kp_g, d_g = ex.extract(img)
        kp_o, d_o = oracle.detect_and_compute(po, img, grid_cell=cell)
        assert kp_g.tobytes() == kp_o.tobytes() and np.array_equal(d_g, d_o), (w, h, cell)
        plain_o, _ = oracle.detect_and_compute(po, img)
        # At most one keypoint per (octave, cell).  (The count is NOT bounded by the plain
        # count: thinning moves the retain-best cut and can admit more ties at the cut score.)
        ws, hs, _, scales = ex.levels()
        lx = np.rint(kp_g["x"] / scales[kp_g["octave"]]).astype(np.int64)
        ly = np.rint(kp_g["y"] / scales[kp_g["octave"]]).astype(np.int64)
        cells = np.stack([kp_g["octave"].astype(np.int64), ly // cell, lx // cell], 1)
        assert len(np.unique(cells, axis=0)) == len(kp_g), (w, h, cell)
        # off again: the reference's selection
        ex.set_grid(0)
        kp_g, d_g = ex.extract(img)
        assert kp_g.tobytes() == plain_o.tobytes()
        with pytest.raises(capi.SlamAmdError):
            ex.set_grid(12)
        ex.close()


@pytest.mark.parametrize("w,h,nf,seed", [(640, 480, 2000, 1), (333, 271, 500, 9), (1241, 376, 4000, 3), (640, 480, 300, 5)])
def test_quadtree_distribution_option_vs_oracle(oracle, capi, w, h, nf, seed):
    """Quadtree distribution with an exact per-level count (an option the reference does not have)
    against its oracle, which walks an explicit tree."""
    img = synth.scene(w, h, seed=seed)
    po = oracle.params(nfeatures=nf)
    ex = capi.Extractor(capi.orb_params(nfeatures=nf), w, h, max_batch=1, max_keypoints=16384)
    ex.set_quadtree(True)
    kp_g, d_g = ex.extract(img)
    kp_o, d_o = oracle.detect_and_compute(po, img, quadtree=True)
    assert len(kp_g) == len(kp_o)
    assert kp_g.tobytes() == kp_o.tobytes() and np.array_equal(d_g, d_o)
    # exact counts: a level never keeps more than its quota, and the plain selection (ties at the
    # cut kept) never keeps fewer
    plain_o, _ = oracle.detect_and_compute(po, img)
    _, _, quotas, _ = ex.levels()
    per_level = np.bincount(kp_g["octave"], minlength=len(quotas))
    plain_level = np.bincount(plain_o["octave"], minlength=len(quotas))
    assert (per_level <= quotas).all() and (per_level <= plain_level).all()
    assert (per_level[plain_level >= quotas] == quotas[plain_level >= quotas]).all()
    assert per_level.sum() < plain_level.sum() or (plain_level <= quotas).all()
    with pytest.raises(capi.SlamAmdError):
        ex.set_grid(16)
    # off again: the reference's selection
    ex.set_quadtree(False)
    kp_g, d_g = ex.extract(img)
    assert kp_g.tobytes() == plain_o.tobytes()
    ex.set_grid(16)
    with pytest.raises(capi.SlamAmdError):
        ex.set_quadtree(True)
    ex.close()


def test_outlier_flags_vs_oracle(oracle, capi):
    """Batched isOutlier (src/map/mappoint.cpp:54-101) against the oracle, flag for flag."""
    n = 5000
    K4, Rt_q, Rt_t, cq, ct, xyz = synth.two_view_setup(n, 11)
    rng = np.random.default_rng(12)
    dist = np.array([0.05, -0.02, 0.001, -0.0005, 0.003])
    uvq = oracle.project_points(Rt_q, K4, dist, xyz); uvt = oracle.project_points(Rt_t, K4, None, xyz)
    kq = np.zeros(n, capi.KEYPOINT_DTYPE); kt = np.zeros(n, capi.KEYPOINT_DTYPE)
    # keypoints scattered around the exact projections: errors on both sides of one pixel
    kq["x"] = uvq[:, 0] + rng.normal(0, 0.6, n); kq["y"] = uvq[:, 1] + rng.normal(0, 0.6, n)
    perm = rng.permutation(n)
    kt["x"][perm] = uvt[:, 0] + rng.normal(0, 0.6, n); kt["y"][perm] = uvt[:, 1] + rng.normal(0, 0.6, n)
    m_ = np.zeros(n, capi.DMATCH_DTYPE); m_["queryIdx"] = np.arange(n); m_["trainIdx"] = perm
    pts = xyz.copy()
    pts[rng.integers(0, n, 50), 0] = np.nan
    pts[rng.integers(0, n, 50)] = (0.0, 0.0, 900.0)
    pts[rng.integers(0, n, 20)] = ct
    pts[rng.integers(0, n, 50), 2] *= -1
    m = capi.Matcher()
    got = m.outlier_flags(pts, m_, kq, kt, cq, ct, Rt_q, K4, dist, Rt_t, K4, None)
    ref = oracle.outlier_flags(pts, m_, kq, kt, cq, ct, Rt_q, K4, dist, Rt_t, K4, None)
    assert np.array_equal(got, ref)
    assert 0.2 < ref.mean() < 0.9
    with pytest.raises(capi.SlamAmdError):
        bad = m_.copy(); bad["queryIdx"][7] = -1
        m.outlier_flags(pts, bad, kq, kt, cq, ct, Rt_q, K4, dist, Rt_t, K4, None)
    assert len(m.outlier_flags(pts[:0], m_[:0], kq, kt, cq, ct, Rt_q, K4, dist, Rt_t, K4, None)) == 0
    m.close()


def test_frontend_grey_resize_vs_oracle(oracle, capi):
    """cvtColor + resize of main.cpp:124-129 in one kernel, against the two-step oracle."""
    import torch
    rng = np.random.default_rng(21)
    for (w, h, target, ch) in ((1920, 1080, 1080, 3), (1280, 720, 1080, 3), (333, 251, 640, 3), (640, 480, 400, 1)):
        frame = rng.integers(0, 256, (h, w, ch), dtype=np.uint8) if ch == 3 else rng.integers(0, 256, (h, w), dtype=np.uint8)
        # smooth structure as well as noise
        yy, xx = np.mgrid[0:h, 0:w]
        ramp = ((xx * 3 + yy * 2) % 256).astype(np.uint8)
        if ch == 3:
            frame[..., 1] = ramp
        fe = capi.Frontend(w, h, target, channels=ch)
        ref = oracle.frontend(frame, target) if ch == 3 else oracle.resize_linear(frame, np.float32(target) / np.float32(w))
        assert (fe.height, fe.width) == ref.shape
        got = fe.run(frame)
        assert np.array_equal(got, ref), (w, h, target, ch)
        # device path, two frames with padded strides
        B = 2
        sp = w * ch + 5
        src = torch.zeros((B, h, sp), dtype=torch.uint8, device="cuda")
        flat = torch.from_numpy(frame.reshape(h, w * ch)).cuda()
        src[0, :, : w * ch] = flat
        src[1, :, : w * ch] = torch.flip(flat, dims=[0])
        dp = fe.width + 3
        dst = torch.zeros((B, fe.height, dp), dtype=torch.uint8, device="cuda")
        fe.run_async(src, B, h * sp, sp, dst, fe.height * dp, dp, torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        out = dst.cpu().numpy()
        assert np.array_equal(out[0, :, : fe.width], ref)
        flipped = frame[::-1].copy()
        ref1 = oracle.frontend(flipped, target) if ch == 3 else oracle.resize_linear(flipped, np.float32(target) / np.float32(w))
        assert np.array_equal(out[1, :, : fe.width], ref1)
        assert (out[:, :, fe.width:] == 0).all()
        fe.close()
    with pytest.raises(capi.SlamAmdError):
        capi.Frontend(640, 480, 320, channels=4)

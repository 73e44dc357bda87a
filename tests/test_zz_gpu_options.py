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
        assert len(kp_o) <= len(plain_o)
        # off again: the reference's selection
        ex.set_grid(0)
        kp_g, d_g = ex.extract(img)
        assert kp_g.tobytes() == plain_o.tobytes()
        with pytest.raises(capi.SlamAmdError):
            ex.set_grid(12)
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

"""GPU parity: the HIP extraction path against the CPU oracle, bit for bit."""
import numpy as np
import pytest

from slam_amd import synth

pytestmark = pytest.mark.gpu


def _compare(oracle, capi, img, nfeatures, **kw):
    po = oracle.params(nfeatures=nfeatures, **kw)
    pg = capi.orb_params(nfeatures=nfeatures, **kw)
    h, w = img.shape
    ex = capi.Extractor(pg, w, h, max_batch=1, max_keypoints=max(8 * nfeatures, 4096))
    try:
        kp_g, d_g = ex.extract(img)
        # pyramid first: any later mismatch is then not the resize
        for l, ref in enumerate(oracle.build_pyramid(po, img)):
            got = ex.debug_level(0, l)
            assert np.array_equal(got, ref), "pyramid level %d differs" % l
            assert np.array_equal(ex.debug_level(0, l, blurred=True), oracle.gaussian_blur7(ref)), \
                "blurred level %d differs" % l
    finally:
        ex.close()
    kp_o, d_o = oracle.detect_and_compute(po, img)
    assert len(kp_g) == len(kp_o), (len(kp_g), len(kp_o))
    for f in ("octave", "response", "size", "class_id"):
        assert np.array_equal(kp_g[f], kp_o[f]), f
    assert np.array_equal(kp_g["x"].view(np.uint32), kp_o["x"].view(np.uint32))
    assert np.array_equal(kp_g["y"].view(np.uint32), kp_o["y"].view(np.uint32))
    # angles: bit-exact (stated tolerance of the path: 0 ulp between GPU and oracle)
    assert np.array_equal(kp_g["angle"].view(np.uint32), kp_o["angle"].view(np.uint32))
    assert np.array_equal(d_g, d_o)
    return kp_g, d_g


def test_extract_640x480_2000(oracle, capi):
    img = synth.scene(640, 480, 1)
    kp, d = _compare(oracle, capi, img, 2000)
    assert len(kp) >= 2000


def test_extract_main_cpp_params_odd_size(oracle, capi):
    # 1000 features as in main.cpp:100, a size whose rows are not 4-byte aligned
    img = synth.scene(701, 463, 7)
    _compare(oracle, capi, img, 1000)


def test_extract_kitti_shape(oracle, capi):
    img = synth.scene(1241, 376, 3)
    _compare(oracle, capi, img, 4000)


def test_extract_gradients_that_pass_the_compass_test_in_both_polarities(oracle, capi):
    """A diagonal ramp 7 (x + y) mod 256: nearly every pixel has a darker neighbour three pixels up and left and a
    brighter one three pixels down and right, so the pre-test passes it in both polarities and a tile's candidate
    list does not hold an entry per (pixel, polarity) -- the kernel falls back to one entry per pixel.  Corners come
    from the ramp's wrap lines and from squares pasted on top."""
    y, x = np.mgrid[0:240, 0:320]
    img = ((7 * (x + y)) % 256).astype(np.uint8)
    rng = np.random.default_rng(11)
    for _ in range(60):
        cx, cy, r = rng.integers(40, 280), rng.integers(40, 200), rng.integers(3, 9)
        img[cy - r:cy + r, cx - r:cx + r] = rng.integers(0, 256)
    kp, d = _compare(oracle, capi, img, 1500)
    assert len(kp) > 200


def test_extract_noise(oracle, capi):
    """Uniform noise: close to 2000 candidate entries in every 64 x 32 tile, corners everywhere."""
    img = np.random.default_rng(5).integers(0, 256, size=(200, 264), dtype=np.uint8)
    kp, d = _compare(oracle, capi, img, 3000)
    assert len(kp) > 500

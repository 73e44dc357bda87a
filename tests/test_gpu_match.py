"""GPU parity: gated Hamming matching against the CPU oracle, bit for bit."""
import numpy as np
import pytest

from slam_amd import synth

pytestmark = pytest.mark.gpu


def _sets(nq, nt, seed):
    d2 = synth.random_descriptors(nt, seed)
    rng = np.random.default_rng(seed + 100)
    src = rng.integers(0, nt, size=nq)
    d1 = synth.noisy_copy(d2[src], 25, seed + 1)
    k1 = synth.random_keypoints(nq, seed=seed + 2)
    k2 = synth.random_keypoints(nt, seed=seed + 3)
    # make most true correspondences spatially close so the window gate keeps some
    k1["x"] = k2["x"][src] + rng.normal(0, 30, nq).astype(np.float32)
    k1["y"] = k2["y"][src] + rng.normal(0, 30, nq).astype(np.float32)
    return d1, k1, d2, k2


@pytest.mark.parametrize("area,max_level", [(-1.0, -1), (-1.0, 4), (50.0, 4), (50.0, -1), (-1.0, 3)])
def test_match_gated_vs_oracle(oracle, capi, area, max_level):
    d1, k1, d2, k2 = _sets(1500, 1700, 11)
    m = capi.Matcher()
    got = m.frame_match(d1, k1, d2, k2, max_dist=300.0, area=area, max_level=max_level)
    ref = oracle.match_gated(d1, k1, d2, k2, 300.0, area, max_level)
    assert np.array_equal(got, ref)
    m.close()


def test_match_ids_and_maxdist(oracle, capi):
    d1, k1, d2, k2 = _sets(900, 800, 21)
    ids = np.arange(0, 900, 3, dtype=np.int32)
    m = capi.Matcher()
    got = m.frame_match(d1, k1, d2, k2, ids=ids, max_dist=40.0, area=-1.0, max_level=3)
    ref = oracle.frame_match(d1, k1, d2, k2, ids, 40.0, -1.0, 3)
    assert np.array_equal(got, ref)
    assert len(got) > 0 and got["distance"].max() <= 40
    m.close()

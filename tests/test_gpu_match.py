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


@pytest.mark.parametrize("kernel", [0, 1])
@pytest.mark.parametrize("area,max_level", [(-1.0, -1), (-1.0, 4), (50.0, 4), (50.0, -1), (-1.0, 3)])
def test_match_gated_vs_oracle(oracle, capi, area, max_level, kernel):
    """All five gate combinations of the reference's call sites, full sweep on the matrix cores
    (kernel 0) and with the popcount kernel (kernel 1)."""
    d1, k1, d2, k2 = _sets(1500, 1700, 11)
    m = capi.Matcher()
    m.debug_set_sweep_kernel(kernel)
    m.debug_set_window_mode(1)
    got = m.frame_match(d1, k1, d2, k2, max_dist=300.0, area=area, max_level=max_level)
    ref = oracle.match_gated(d1, k1, d2, k2, 300.0, area, max_level)
    assert np.array_equal(got, ref)
    m.close()


@pytest.mark.parametrize("nq,nt", [(1, 1), (33, 31), (64, 65), (257, 64), (300, 1000), (513, 4097)])
def test_matrix_core_gates_ragged_sizes_and_sorted_octaves(oracle, capi, nq, nt):
    """The gates in the matrix-core kernel: sizes around the 32-row blocks and the 256-query workgroups; octaves
    sorted as an extractor leaves them (whole blocks above or below max_level take the kernel's short cuts) and
    random; train rows tied in distance where the nearer one fails a gate."""
    d1, k1, d2, k2 = _sets(nq, nt, 300 + nq + nt)
    if nt >= 64:
        d2[40] = d2[7]                       # equal rows: the lower index wins unless a gate removes it
        k2["octave"][7] = 7
        k2["x"][40] = k2["x"][7] + 500.0
    if nq >= 4:
        d1[1] = d2[min(7, nt - 1)]
        k1["octave"][1] = 7                  # high query: row 7 is gated out, row 40 must win
        d1[2] = d2[min(7, nt - 1)]
        k1["octave"][2] = 0
    m = capi.Matcher()
    m.debug_set_window_mode(1)
    for sort_octaves in (False, True):
        if sort_octaves:
            k1["octave"] = np.sort(k1["octave"]); k2["octave"] = np.sort(k2["octave"])
        for area, lvl in ((-1.0, 4), (45.0, 4), (45.0, -1), (-1.0, 0), (-1.0, 7), (0.0, -1), (float("inf"), 2)):
            ref = oracle.match_gated(d1, k1, d2, k2, 300.0, area, lvl)
            got = m.frame_match(d1, k1, d2, k2, max_dist=300.0, area=area, max_level=lvl)
            assert np.array_equal(got, ref), (area, lvl, sort_octaves)
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


def _rotated_angles(k1, k2, src, seed, turn=37.0, noise=4.0, outliers=0.2):
    """Query angles = matched train angle + a common turn + noise, a share of them random."""
    rng = np.random.default_rng(seed)
    a = k2["angle"][src] + turn + rng.normal(0, noise, len(k1))
    bad = rng.random(len(k1)) < outliers
    a[bad] = rng.uniform(0, 360, bad.sum())
    k1["angle"] = np.mod(a, 360.0).astype(np.float32)


@pytest.mark.parametrize("ratio,histo", [(0.7, 0), (0.95, 0), (0.0, 30), (0.8, 30), (0.9, 12), (1.5, 64)])
@pytest.mark.parametrize("area,max_level", [(-1.0, -1), (60.0, 4)])
def test_match_options_vs_oracle(oracle, capi, ratio, histo, area, max_level):
    nq, nt, seed = 1300, 1900, 31
    d1, k1, d2, k2 = _sets(nq, nt, seed)
    src = np.random.default_rng(seed + 100).integers(0, nt, size=nq)
    _rotated_angles(k1, k2, src, seed + 5)
    # near-duplicate train rows, so that the ratio test has something to reject
    d2[1::7] = synth.noisy_copy(d2[0::7][: len(d2[1::7])], 6, seed + 9)
    m = capi.Matcher()
    m.set_options(ratio, histo)
    got = m.frame_match(d1, k1, d2, k2, max_dist=80.0, area=area, max_level=max_level)
    ref = oracle.match_gated_ex(d1, k1, d2, k2, 80.0, area, max_level, ratio, histo)
    plain = oracle.match_gated(d1, k1, d2, k2, 80.0, area, max_level)
    assert np.array_equal(got, ref)
    assert 0 < len(ref) <= len(plain)
    if ratio in (0.7, 0.8) or histo in (30, 12):
        assert len(ref) < len(plain)
    # switching the options off restores the reference's behaviour
    m.set_options(0.0, 0)
    assert np.array_equal(m.frame_match(d1, k1, d2, k2, max_dist=80.0, area=area, max_level=max_level), plain)
    m.close()


def test_match_options_ids_and_errors(oracle, capi):
    d1, k1, d2, k2 = _sets(700, 650, 41)
    ids = np.arange(1, 700, 2, dtype=np.int32)
    m = capi.Matcher()
    m.set_options(0.85, 30)
    got = m.frame_match(d1, k1, d2, k2, ids=ids, max_dist=300.0, area=-1.0, max_level=4)
    ref = oracle.match_gated_ex(d1[ids], k1[ids], d2, k2, 300.0, -1.0, 4, 0.85, 30)
    ref["queryIdx"] = ids[ref["queryIdx"]]
    assert np.array_equal(got, ref)
    with pytest.raises(capi.SlamAmdError):
        m.set_options(-1.0, 0)
    with pytest.raises(capi.SlamAmdError):
        m.set_options(0.5, 65)
    m.close()


@pytest.mark.parametrize("nq,nt,area,max_level", [(1500, 1700, 50.0, 4), (1500, 1700, 12.5, -1), (70, 3000, 200.0, 3),
                                                   (5000, 63, 30.0, 4), (1, 1, 5.0, -1), (2500, 2500, 1e9, 4)])
def test_cell_sorted_window_match_equals_full_sweep(oracle, capi, nq, nt, area, max_level):
    d1, k1, d2, k2 = _sets(nq, nt, 51)
    ref = oracle.match_gated(d1, k1, d2, k2, 300.0, area, max_level)
    m = capi.Matcher()
    for mode in (3, 2, 1, 0):
        m.debug_set_window_mode(mode)
        got = m.frame_match(d1, k1, d2, k2, max_dist=300.0, area=area, max_level=max_level)
        assert np.array_equal(got, ref), mode
    m.close()


@pytest.mark.parametrize("nt", [2030, 8192, 700, 12000, 32768])
def test_cell_sorted_matrix_core_match_ties_and_options(oracle, capi, nt):
    """Reordered sets on the matrix cores: equal train rows at nearby positions (the lowest original index must
    win wherever the reordering puts the rows), the ratio test and the rotation check, ids and query positions."""
    nq = 2030
    d1, k1, d2, k2 = _sets(nq, nt, 91 + nt)
    rng = np.random.default_rng(nt)
    # groups of identical train rows scattered over the set, all inside one window
    for g in range(40):
        rows = rng.choice(nt, size=5, replace=False)
        d2[rows] = d2[rows[0]]
        k2["x"][rows] = k2["x"][rows[0]] + rng.uniform(-20, 20, 5).astype(np.float32)
        k2["y"][rows] = k2["y"][rows[0]] + rng.uniform(-20, 20, 5).astype(np.float32)
        q = rng.integers(0, nq)
        d1[q] = d2[rows[0]]
        k1["x"][q] = k2["x"][rows[0]]; k1["y"][q] = k2["y"][rows[0]]
    m = capi.Matcher()
    m.debug_set_window_mode(3)
    for area, lvl in ((50.0, 4), (50.0, -1), (8.0, 2), (1e9, 4)):
        got = m.frame_match(d1, k1, d2, k2, max_dist=300.0, area=area, max_level=lvl)
        ref = oracle.match_gated(d1, k1, d2, k2, 300.0, area, lvl)
        assert np.array_equal(got, ref), (area, lvl)
    ids = np.sort(rng.permutation(nq)[:1500]).astype(np.int32)
    qpos = np.stack([k1["x"][ids], k1["y"][ids]], 1).astype(np.float32) + np.float32(3.0)
    kq = k1[ids].copy(); kq["x"] = qpos[:, 0]; kq["y"] = qpos[:, 1]
    ref = oracle.match_gated(d1[ids], kq, d2, k2, 300.0, 50.0, 4)
    ref["queryIdx"] = ids[ref["queryIdx"]]
    got = m.frame_match(d1, k1, d2, k2, ids=ids, q_xy=qpos, max_dist=300.0, area=50.0, max_level=4)
    assert np.array_equal(got, ref)
    m.set_options(0.9, 30)
    got = m.frame_match(d1, k1, d2, k2, max_dist=300.0, area=45.0, max_level=4)
    assert np.array_equal(got, oracle.match_gated_ex(d1, k1, d2, k2, 300.0, 45.0, 4, 0.9, 30))
    m.close()


def test_cell_sorted_window_match_odd_positions(oracle, capi):
    """Positions that are NaN, infinite, huge, negative or all equal; ids and query positions."""
    nq, nt = 900, 1100
    d1, k1, d2, k2 = _sets(nq, nt, 61)
    rng = np.random.default_rng(5)
    k2["x"][rng.integers(0, nt, 20)] = np.nan
    k2["y"][rng.integers(0, nt, 20)] = np.inf
    k2["x"][rng.integers(0, nt, 20)] = -np.inf
    k2["x"][:30] = 3.0e38
    k2["y"][:30] = -3.0e38
    q_xy = np.stack([k1["x"], k1["y"]], 1).astype(np.float32)
    q_xy[rng.integers(0, nq, 30)] = np.nan
    q_xy[rng.integers(0, nq, 30), 0] = np.inf
    q_xy[:15] = (3.0e38, -3.0e38)          # coincide with the huge train positions: they must still match
    q_xy[15:40] = (-250.0, -1000.5)
    ids = rng.permutation(nq)[:700].astype(np.int32)
    ids.sort()
    qpos = q_xy[:len(ids)]
    kq = k1[ids].copy(); kq["x"] = qpos[:, 0]; kq["y"] = qpos[:, 1]
    m = capi.Matcher()
    for area in (40.0, 1.0e30):
        ref = oracle.match_gated(d1[ids], kq, d2, k2, 300.0, area, 4)
        ref["queryIdx"] = ids[ref["queryIdx"]]
        for mode in (3, 2, 1):
            m.debug_set_window_mode(mode)
            got = m.frame_match(d1, k1, d2, k2, ids=ids, q_xy=qpos, max_dist=300.0, area=area, max_level=4)
            assert np.array_equal(got, ref), (area, mode)
    # every position identical
    k1["x"] = k1["y"] = 7.0; k2["x"] = k2["y"] = 7.25
    for mode in (3, 2):
        m.debug_set_window_mode(mode)
        got = m.frame_match(d1, k1, d2, k2, max_dist=300.0, area=0.5, max_level=-1)
        assert np.array_equal(got, oracle.match_gated(d1, k1, d2, k2, 300.0, 0.5, -1)) and len(got) == nq
    # the options ride on the cell-sorted path as well
    d1, k1, d2, k2 = _sets(nq, nt, 62)
    m.set_options(0.9, 30)
    got = m.frame_match(d1, k1, d2, k2, max_dist=300.0, area=45.0, max_level=4)
    assert np.array_equal(got, oracle.match_gated_ex(d1, k1, d2, k2, 300.0, 45.0, 4, 0.9, 30))
    m.close()


def test_match_10000_x_10000_vs_oracle(oracle, capi):
    """The 4K stream configuration's matching size: 10 000 x 10 000 rows, gates off (brute force), with the
    octave gate, and with the 50-pixel window (cell-sorted path), against the multi-threaded oracle."""
    n = 10000
    d2 = synth.random_descriptors(n, 31)
    rng = np.random.default_rng(32)
    src = rng.integers(0, n, size=n)
    d1 = synth.noisy_copy(d2[src], 30, 33)
    k1 = synth.random_keypoints(n, 3840, 2160, seed=34)
    k2 = synth.random_keypoints(n, 3840, 2160, seed=35)
    k1["x"] = k2["x"][src] + rng.normal(0, 25, n).astype(np.float32)
    k1["y"] = k2["y"][src] + rng.normal(0, 25, n).astype(np.float32)
    m = capi.Matcher()
    for area, lvl in ((-1.0, -1), (-1.0, 4), (50.0, 4)):
        ref = oracle.match_gated_mt(d1, k1, d2, k2, 300.0, area, lvl)
        for mode in (0, 1):                  # by size (cell-sorted where the window is on), then the full sweep
            m.debug_set_window_mode(mode)
            got = m.frame_match(d1, k1, d2, k2, max_dist=300.0, area=area, max_level=lvl)
            assert got.tobytes() == ref.tobytes() and len(got) > 5000, (area, lvl, mode, len(got), len(ref))
    m.close()


@pytest.mark.parametrize("nq,nt", [(1, 1), (1, 33), (31, 32), (33, 31), (257, 64), (300, 1000), (2000, 2000), (513, 4097)])
def test_matrix_core_match_equals_oracle_and_popcount_kernel(oracle, capi, nq, nt):
    """Gates off: distances come from the i8 matrix-core product; ragged sizes around the 32-row
    tiles and the 256-query workgroups, ties (duplicated train rows) and all-zero / all-one rows."""
    d1, k1, d2, k2 = _sets(nq, nt, 71 + nq + nt)
    if nt >= 8:
        d2[3] = d2[5]            # equal distances: the lower train index must win
        d2[nt - 1] = 0
        d2[nt - 2] = 255
    if nq >= 4:
        d1[0] = 0
        d1[1] = 255
        d1[2] = d2[min(5, nt - 1)]
    if nt >= 128 and nq >= 8:
        # equal rows in different 32-row blocks, the later one at a lower row of its block,
        # and in different lane halves of one block
        d2[66] = d2[37]
        d2[97] = d2[37]
        d2[12] = d2[9]
        d1[5] = d2[37]
        d1[6] = d2[9]
        d1[7] = d2[97] ^ np.uint8(1)
    ref = oracle.match_gated(d1, k1, d2, k2, 300.0, -1.0, -1)
    if nt >= 128 and nq >= 8:
        assert ref[5]["trainIdx"] == 37 and ref[6]["trainIdx"] == 9 and ref[7]["trainIdx"] == 37
    m = capi.Matcher()
    got = m.frame_match(d1, k1, d2, k2, max_dist=300.0, area=-1.0, max_level=-1)
    m.debug_set_sweep_kernel(1)
    got_valu = m.frame_match(d1, k1, d2, k2, max_dist=300.0, area=-1.0, max_level=-1)
    m.debug_set_sweep_kernel(0)
    got_lvl = m.frame_match(d1, k1, d2, k2, max_dist=300.0, area=-1.0, max_level=2)
    assert np.array_equal(got_lvl, oracle.match_gated(d1, k1, d2, k2, 300.0, -1.0, 2))
    ids = np.arange(nq - 1, -1, -2, dtype=np.int32)
    got_ids = m.frame_match(d1, k1, d2, k2, ids=ids, max_dist=60.0, area=-1.0, max_level=-1)
    ref_ids = oracle.frame_match(d1, k1, d2, k2, ids, 60.0, -1.0, -1)
    m.close()
    assert np.array_equal(got, ref)
    assert np.array_equal(got_valu, ref)
    assert np.array_equal(got_ids, ref_ids)

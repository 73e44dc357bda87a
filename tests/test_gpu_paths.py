"""GPU parity of the device-pointer entry points, edge cases and full-size properties."""
import os
import sys

import numpy as np
import pytest

from slam_amd import synth

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
KP, DM = 28, 16


def _torch():
    import torch
    return torch


def _dev_kp(t, arr):
    return t.from_numpy(np.ascontiguousarray(arr).view(np.uint8).copy()).cuda()


def _kp_from(t_u8, n):
    from slam_amd import capi
    return t_u8.cpu().numpy().view(capi.KEYPOINT_DTYPE)[:n].copy()


# ---------------------------------------------------------------- extraction ----
def test_extract_batch_async_with_padding_and_offset(oracle, capi):
    """Three different frames in one call, rows padded, base pointer not 4-byte aligned."""
    torch = _torch()
    W, H, NF, CAP, B = 333, 251, 600, 2048, 3
    frames = [synth.scene(W, H, 20 + i) for i in range(B)]
    for stride, offset in ((W, 0), (W + 7, 0), (352, 1), (352, 64)):
        pitch = stride * H + 13
        buf = np.zeros(offset + B * pitch, np.uint8)
        for b in range(B):
            view = buf[offset + b * pitch: offset + b * pitch + stride * H].reshape(H, stride)
            view[:, :W] = frames[b]
            view[:, W:] = 255 - (b * 40)          # padding content must not matter
        d_img = torch.from_numpy(buf).cuda()
        d_kp = torch.zeros(B * CAP * KP, dtype=torch.uint8, device="cuda")
        d_desc = torch.zeros(B * CAP * 32, dtype=torch.uint8, device="cuda")
        d_cnt = torch.zeros(B, dtype=torch.int32, device="cuda")
        ex = capi.Extractor(capi.orb_params(nfeatures=NF), W, H, max_batch=B, max_keypoints=CAP)
        ex.extract_batch_async(d_img.data_ptr() + offset, B, pitch, stride, d_kp, d_desc, d_cnt, CAP,
                               torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        cnt = d_cnt.cpu().numpy()
        po = oracle.params(nfeatures=NF)
        for b in range(B):
            kp_o, d_o = oracle.detect_and_compute(po, frames[b])
            assert cnt[b] == len(kp_o)
            kp_g = d_kp.cpu().numpy().view(capi.KEYPOINT_DTYPE)[b * CAP: b * CAP + cnt[b]]
            d_g = d_desc.cpu().numpy().reshape(B * CAP, 32)[b * CAP: b * CAP + cnt[b]]
            assert kp_g.tobytes() == kp_o.tobytes(), (stride, offset, b)
            assert np.array_equal(d_g, d_o), (stride, offset, b)
        ex.close()


def test_extract_batches_of_8_and_other_sizes_on_one_extractor(oracle, capi):
    """A multiple of 8 frames runs the corner kernel on a grid of its own (whole frames per XCD); the same
    extractor then takes 5 and 16 frames: every frame of every call against the oracle."""
    torch = _torch()
    W, H, NF, CAP, MAXB = 200, 152, 300, 1024, 16
    frames = [synth.scene(W, H, 70 + i) for i in range(MAXB)]
    po = oracle.params(nfeatures=NF)
    want = [oracle.detect_and_compute(po, f) for f in frames]
    ex = capi.Extractor(capi.orb_params(nfeatures=NF), W, H, max_batch=MAXB, max_keypoints=CAP)
    d_img = torch.from_numpy(np.stack(frames)).cuda()
    for B in (8, 5, 16):
        d_kp = torch.zeros(B * CAP * KP, dtype=torch.uint8, device="cuda")
        d_desc = torch.zeros(B * CAP * 32, dtype=torch.uint8, device="cuda")
        d_cnt = torch.zeros(B, dtype=torch.int32, device="cuda")
        ex.extract_batch_async(d_img.data_ptr(), B, W * H, W, d_kp, d_desc, d_cnt, CAP, torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        cnt = d_cnt.cpu().numpy()
        kp_all = d_kp.cpu().numpy().view(capi.KEYPOINT_DTYPE)
        d_all = d_desc.cpu().numpy().reshape(B * CAP, 32)
        for b in range(B):
            kp_o, d_o = want[b]
            assert cnt[b] == len(kp_o), (B, b)
            assert kp_all[b * CAP: b * CAP + cnt[b]].tobytes() == kp_o.tobytes(), (B, b)
            assert np.array_equal(d_all[b * CAP: b * CAP + cnt[b]], d_o), (B, b)
    ex.close()


def test_extract_capacity_and_errors(oracle, capi):
    torch = _torch()
    img = synth.scene(320, 240, 31)
    po = oracle.params(nfeatures=400)
    kp_o, d_o = oracle.detect_and_compute(po, img)
    n = len(kp_o)
    ex = capi.Extractor(capi.orb_params(nfeatures=400), 320, 240, max_batch=1, max_keypoints=2048)
    with pytest.raises(capi.SlamAmdError) as e:          # caller buffer too small: nothing written
        ex.extract(img, capacity=n - 1)
    assert e.value.status == capi.ERR_CAPACITY
    kp, d = ex.extract_fetch(n)                           # ... and fetched afterwards without a second extraction
    assert kp.tobytes() == kp_o.tobytes() and np.array_equal(d, d_o)
    with pytest.raises(capi.SlamAmdError):
        ex.extract_fetch(n - 1)
    kp, d = ex.extract(img, capacity=n)                   # exactly enough
    assert kp.tobytes() == kp_o.tobytes() and np.array_equal(d, d_o)
    with pytest.raises(capi.SlamAmdError) as e:          # wrong image size
        ex.extract(np.zeros((100, 100), np.uint8))
    assert e.value.status == capi.ERR_ARG
    # device path with a small capacity: true count reported, first `cap` keypoints written
    cap = 100
    d_img = torch.from_numpy(img).cuda()
    d_kp = torch.zeros(cap * KP, dtype=torch.uint8, device="cuda")
    d_desc = torch.zeros(cap * 32, dtype=torch.uint8, device="cuda")
    d_cnt = torch.zeros(1, dtype=torch.int32, device="cuda")
    ex.extract_batch_async(d_img, 1, 0, 320, d_kp, d_desc, d_cnt, cap, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    assert int(d_cnt.item()) == n
    assert _kp_from(d_kp, cap).tobytes() == kp_o[:cap].tobytes()
    assert np.array_equal(d_desc.cpu().numpy().reshape(cap, 32), d_o[:cap])
    ex.close()


def test_extract_degenerate_images(oracle, capi):
    ex = capi.Extractor(capi.orb_params(nfeatures=100), 160, 120, max_batch=1, max_keypoints=512)
    kp, d = ex.extract(np.full((120, 160), 90, np.uint8))            # flat: no corners
    assert len(kp) == 0 and d.shape == (0, 32)
    kp, d = ex.extract(np.zeros((0, 0), np.uint8))                    # empty image -> empty result
    assert len(kp) == 0
    ex.close()
    img = synth.scene(70, 66, 3)                                      # levels >= 1 are <= 62 px
    ex = capi.Extractor(capi.orb_params(nfeatures=100), 70, 66, max_batch=1, max_keypoints=512)
    kp, d = ex.extract(img)
    kp_o, d_o = oracle.detect_and_compute(oracle.params(nfeatures=100), img)
    assert kp.tobytes() == kp_o.tobytes() and np.array_equal(d, d_o) and (kp["octave"] == 0).all()
    ex.close()
    ex = capi.Extractor(capi.orb_params(nfeatures=0), 160, 120, max_batch=1, max_keypoints=512)
    assert len(ex.extract(synth.scene(160, 120, 3))[0]) == 0         # nfeatures 0 keeps nothing
    ex.close()


def test_extract_select_without_cache_ranks_from_memory(oracle, capi):
    """Levels that keep more corners than the selection step's LDS cache holds are ranked from memory."""
    img = synth.scene(400, 300, 43)
    kp_o, d_o = oracle.detect_and_compute(oracle.params(nfeatures=800), img)
    for entries in (0, 1, 100, 4096):
        ex = capi.Extractor(capi.orb_params(nfeatures=800), 400, 300, max_batch=1, max_keypoints=4096)
        ex.debug_set_select_cache(entries)
        kp, d = ex.extract(img)
        ex.close()
        assert kp.tobytes() == kp_o.tobytes() and np.array_equal(d, d_o), entries


def test_extract_other_parameters(oracle, capi):
    img = synth.scene(400, 300, 41)
    for kw in (dict(nfeatures=700, scale_factor=1.5, nlevels=5), dict(nfeatures=300, scale_factor=2.0, nlevels=3),
               dict(nfeatures=500, fast_threshold=35, edge_threshold=40), dict(nfeatures=900, nlevels=1)):
        ex = capi.Extractor(capi.orb_params(**kw), 400, 300, max_batch=1, max_keypoints=4096)
        kp, d = ex.extract(img)
        po = oracle.params(**kw)
        for l, ref in enumerate(oracle.build_pyramid(po, img)):
            assert np.array_equal(ex.debug_level(0, l), ref), (kw, l)
        ex.close()
        kp_o, d_o = oracle.detect_and_compute(po, img)
        assert kp.tobytes() == kp_o.tobytes() and np.array_equal(d, d_o), kw


def test_extract_fast_threshold_sweep(oracle, capi):
    """Every remainder of the threshold in the packed pre-test's rounding ((t + 1) / 4), both ends of its range, and
    images on which nearly every pixel is a candidate (threshold 0 on noise: the candidate list overflows and is
    rebuilt with one entry per pixel) or none is."""
    rng = np.random.default_rng(77)
    noise = rng.integers(0, 256, (150, 200), dtype=np.uint8)
    scene = synth.scene(200, 150, 44)
    for img in (noise, scene):
        for t in (0, 1, 2, 3, 4, 6, 19, 50, 127, 254, 255):
            kw = dict(nfeatures=300, fast_threshold=t)
            ex = capi.Extractor(capi.orb_params(**kw), 200, 150, max_batch=1, max_keypoints=4096)
            kp, d = ex.extract(img)
            ex.close()
            kp_o, d_o = oracle.detect_and_compute(oracle.params(**kw), img)
            assert kp.tobytes() == kp_o.tobytes() and np.array_equal(d, d_o), t


def test_extract_chessboard_golden(oracle, capi):
    """The reference's own image asset: many equal scores, so ties at the cut are exercised."""
    sys.path.insert(0, os.path.join(HERE, "golden"))
    import make_golden
    g = np.load(os.path.join(HERE, "golden", "chessboard_orb.npz"))
    a, b = make_golden.chessboard_pair()
    h, w = a.shape
    ex = capi.Extractor(capi.orb_params(nfeatures=1000), w, h, max_batch=1, max_keypoints=8192)
    kp_a, d_a = ex.extract(a)
    kp_b, d_b = ex.extract(b)
    ex.close()
    assert kp_a.tobytes() == g["kp_a"].tobytes() and np.array_equal(d_a, g["desc_a"])
    assert kp_b.tobytes() == g["kp_b"].tobytes() and np.array_equal(d_b, g["desc_b"])
    m = capi.Matcher()
    got = m.frame_match(d_a, kp_a, d_b, kp_b, max_dist=300.0, area=-1.0, max_level=4)   # mapper.cpp:34
    m.close()
    assert got.tobytes() == g["matches"].tobytes()


def test_extract_4k_10000_features(oracle, capi):
    img = synth.scene(3840, 2160, 5)
    ex = capi.Extractor(capi.orb_params(nfeatures=10000), 3840, 2160, max_batch=1, max_keypoints=16384)
    kp, d = ex.extract(img)
    ex.close()
    kp_o, d_o = oracle.detect_and_compute(oracle.params(nfeatures=10000), img)
    assert len(kp) == len(kp_o) >= 10000
    assert kp.tobytes() == kp_o.tobytes() and np.array_equal(d, d_o)


# ------------------------------------------------------------------ matching ----
def _sets(nq, nt, seed, w=640, h=480):
    d2 = synth.random_descriptors(nt, seed)
    rng = np.random.default_rng(seed + 100)
    src = rng.integers(0, nt, size=nq)
    d1 = synth.noisy_copy(d2[src], 25, seed + 1)
    k1 = synth.random_keypoints(nq, w, h, seed=seed + 2)
    k2 = synth.random_keypoints(nt, w, h, seed=seed + 3)
    k1["x"] = k2["x"][src] + rng.normal(0, 30, nq).astype(np.float32)
    k1["y"] = k2["y"][src] + rng.normal(0, 30, nq).astype(np.float32)
    return d1, k1, d2, k2


def test_match_edge_cases(oracle, capi):
    m = capi.Matcher()
    d1, k1, d2, k2 = _sets(130, 70, 5)
    assert len(m.frame_match(d1[:0], k1[:0], d2, k2)) == 0                       # no queries
    assert len(m.frame_match(d1, k1, d2[:0], k2[:0])) == 0                       # no train rows
    for nq, nt in ((1, 1), (1, 70), (130, 1), (65, 63), (129, 64), (130, 70)):   # ragged sizes
        got = m.frame_match(d1[:nq], k1[:nq], d2[:nt], k2[:nt], area=60.0, max_level=3)
        assert got.tobytes() == oracle.match_gated(d1[:nq], k1[:nq], d2[:nt], k2[:nt], 300.0, 60.0, 3).tobytes()
    # ties: identical train rows -> lowest index; nothing within max_dist -> no match
    dt = np.repeat(d1[:1], 50, axis=0)
    got = m.frame_match(d1[:4], k1[:4], dt, k2[:50], area=-1.0, max_level=-1)
    assert got["trainIdx"].tolist() == [0, 0, 0, 0]
    assert got.tobytes() == oracle.match_gated(d1[:4], k1[:4], dt, k2[:50], 300.0, -1.0, -1).tobytes()
    far = (255 - d1[:1])
    assert len(m.frame_match(d1[:1], k1[:1], far, k2[:1], max_dist=255.0, area=-1.0, max_level=-1)) == 0
    assert len(m.frame_match(d1[:1], k1[:1], far, k2[:1], max_dist=256.0, area=-1.0, max_level=-1)) == 1
    assert len(m.frame_match(d1[:1], k1[:1], far, k2[:1], max_dist=-1.0, area=-1.0, max_level=-1)) == 0
    # gates off: keypoints may be omitted
    got = m.frame_match(d1, None, d2, None, area=-1.0, max_level=-1)
    assert got.tobytes() == oracle.match_gated(d1, k1, d2, k2, 300.0, -1.0, -1).tobytes()
    with pytest.raises(capi.SlamAmdError):
        m.frame_match(d1, None, d2, None, area=50.0, max_level=-1)
    with pytest.raises(capi.SlamAmdError):
        m.frame_match(d1, k1, d2, k2, ids=[0, 500])
    m.close()


def test_projection_windowed_match_20k_vs_4k(oracle, capi):
    """KITTI-shaped benchmark configuration: 20000 map points with projected positions vs 4000 keypoints."""
    d1, k1, d2, k2 = _sets(20000, 4000, 7, 1241, 376)
    rng = np.random.default_rng(9)
    q_xy = np.stack([k1["x"] + rng.normal(0, 5, 20000), k1["y"] + rng.normal(0, 5, 20000)], 1).astype(np.float32)
    ids = np.arange(20000, dtype=np.int32)
    m = capi.Matcher()
    got = m.frame_match(d1, k1, d2, k2, ids=ids, q_xy=q_xy, max_dist=300.0, area=50.0, max_level=4)
    m.debug_set_window_mode(1)
    full = m.frame_match(d1, k1, d2, k2, ids=ids, q_xy=q_xy, max_dist=300.0, area=50.0, max_level=4)
    m.close()
    kq = k1.copy(); kq["x"] = q_xy[:, 0]; kq["y"] = q_xy[:, 1]
    ref = oracle.match_gated(d1, kq, d2, k2, 300.0, 50.0, 4)
    assert got.tobytes() == ref.tobytes() and len(got) > 10000
    assert full.tobytes() == ref.tobytes()


def test_frame_match_chain(oracle, capi):
    """Every frame of a batch against the one before it, frame 0 against a frame held elsewhere: the
    chained call against the oracle's match of each consecutive pair (matrix-core path, gated popcount
    path, and the ratio / rotation options)."""
    torch = _torch()
    st = torch.cuda.current_stream().cuda_stream
    B, CAP = 5, 384
    counts = [300, 384, 0, 77, 201]
    n_prev = 250
    sets = [_sets(CAP, CAP, 90 + b) for b in range(B + 1)]
    d_all = np.concatenate([sets[b][0] for b in range(B)]); k_all = np.concatenate([sets[b][1] for b in range(B)])
    D = torch.from_numpy(d_all).cuda(); K = _dev_kp(torch, k_all)
    Dp = torch.from_numpy(sets[B][0]).cuda(); Kp = _dev_kp(torch, sets[B][1])
    N = torch.tensor(counts, dtype=torch.int32, device="cuda"); Np = torch.tensor([n_prev], dtype=torch.int32, device="cuda")
    out = torch.zeros(B * CAP * DM, dtype=torch.uint8, device="cuda")
    nout = torch.zeros(B, dtype=torch.int32, device="cuda")
    m = capi.Matcher()
    m.reserve(B, CAP, CAP)

    def sides(b):
        if b == 0:
            return sets[B][0][:n_prev], sets[B][1][:n_prev]
        return sets[b - 1][0][:counts[b - 1]], sets[b - 1][1][:counts[b - 1]]

    for ratio, histo in ((0.0, 0), (0.9, 30)):
        m.set_options(ratio, histo)
        m.reserve(B, CAP, CAP)
        for area, lvl, mode in ((-1.0, -1, 0), (80.0, 4, 3), (25.0, -1, 3), (80.0, 4, 1), (25.0, -1, 1)):
            m.debug_set_window_mode(mode)       # 3: both sets reordered by grid cell, 1: as they are
            m.frame_match_chain_async(B, Dp, Kp, Np, D, K, CAP, N, out, nout, 300.0, area, lvl, st)
            torch.cuda.synchronize()
            res = out.cpu().numpy().view(capi.DMATCH_DTYPE)
            for b in range(B):
                d1, k1 = sides(b)
                d2, k2 = sets[b][0][:counts[b]], sets[b][1][:counts[b]]
                if ratio > 0:
                    ref = oracle.match_gated_ex(d1, k1, d2, k2, 300.0, area, lvl, ratio, histo)
                else:
                    ref = oracle.match_gated(d1, k1, d2, k2, 300.0, area, lvl)
                assert int(nout[b].item()) == len(ref), (b, area, ratio)
                assert res[b * CAP: b * CAP + len(ref)].tobytes() == ref.tobytes(), (b, area, ratio)


def test_windowed_chain_under_graph_capture(oracle, capi):
    """The window-gated chained match (both sets reordered by grid cell in the matcher's scratch) recorded into a
    graph: refused while the scratch is not reserved, and two replays over different inputs equal the oracle."""
    torch = _torch()
    B, CAP = 3, 640
    counts = [600, 640, 333]
    def load(seed):
        sets = [_sets(CAP, CAP, seed + b) for b in range(B + 1)]
        d = np.concatenate([sets[b][0] for b in range(B)]); k = np.concatenate([sets[b][1] for b in range(B)])
        return sets, torch.from_numpy(d).cuda(), _dev_kp(torch, k), torch.from_numpy(sets[B][0]).cuda(), _dev_kp(torch, sets[B][1])
    sets, D, K, Dp, Kp = load(300)
    N = torch.tensor(counts, dtype=torch.int32, device="cuda"); Np = torch.tensor([500], dtype=torch.int32, device="cuda")
    out = torch.zeros(B * CAP * DM, dtype=torch.uint8, device="cuda")
    nout = torch.zeros(B, dtype=torch.int32, device="cuda")
    m = capi.Matcher()
    m.reserve(B, CAP, 0)            # keys only: nothing for the reordered sets
    g = torch.cuda.CUDAGraph()
    refused = False
    try:
        with torch.cuda.graph(g):
            m.frame_match_chain_async(B, Dp, Kp, Np, D, K, CAP, N, out, nout, 300.0, 60.0, 4,
                                      torch.cuda.current_stream().cuda_stream)
    except capi.SlamAmdError:
        refused = True
    assert refused
    del g
    m.close()
    m = capi.Matcher()
    m.reserve(B, CAP, CAP)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        m.frame_match_chain_async(B, Dp, Kp, Np, D, K, CAP, N, out, nout, 300.0, 60.0, 4,
                                  torch.cuda.current_stream().cuda_stream)
    for seed in (300, 400):
        if seed != 300:
            sets, D2, K2, Dp2, Kp2 = load(seed)
            D.copy_(D2); K.copy_(K2); Dp.copy_(Dp2); Kp.copy_(Kp2)
        out.zero_(); nout.zero_()
        g.replay()
        torch.cuda.synchronize()
        res = out.cpu().numpy().view(capi.DMATCH_DTYPE)
        for b in range(B):
            d1, k1 = (sets[B][0][:500], sets[B][1][:500]) if b == 0 else (sets[b - 1][0][:counts[b - 1]], sets[b - 1][1][:counts[b - 1]])
            ref = oracle.match_gated(d1, k1, sets[b][0][:counts[b]], sets[b][1][:counts[b]], 300.0, 60.0, 4)
            assert int(nout[b].item()) == len(ref), (seed, b)
            assert res[b * CAP: b * CAP + len(ref)].tobytes() == ref.tobytes(), (seed, b)
    del g
    m.close()


def test_windowed_chain_on_extracted_frames_all_paths_agree(oracle, capi):
    """Benchmark-shaped input (640 x 480 frames through the extractor, ~2000 keypoints each, clustered as real corners
    are): the chained match with the window and octave gates through the reordered matrix-core path, the gated full
    sweep and the popcount kernel gives the same bytes; one frame pair is also checked against the oracle."""
    torch = _torch()
    st = torch.cuda.current_stream().cuda_stream
    W, H, B, CAP = 640, 480, 12, 4096
    ex = capi.Extractor(capi.orb_params(nfeatures=2000), W, H, max_batch=B, max_keypoints=CAP)
    frames = synth.frame_sequence(W, H, B, seed=21)
    d_img = torch.from_numpy(np.stack(frames)).cuda()
    kp = torch.zeros(B * CAP * KP, dtype=torch.uint8, device="cuda")
    desc = torch.zeros(B * CAP * 32, dtype=torch.uint8, device="cuda")
    cnt = torch.zeros(B, dtype=torch.int32, device="cuda")
    ex.extract_batch_async(d_img, B, H * W, W, kp.data_ptr(), desc.data_ptr(), cnt.data_ptr(), CAP, st)
    torch.cuda.synchronize()
    counts = cnt.cpu().numpy()
    assert counts.min() > 1500
    m = capi.Matcher()
    m.reserve(B, CAP, CAP)
    out = torch.zeros(B * CAP * DM, dtype=torch.uint8, device="cuda")
    nout = torch.zeros(B, dtype=torch.int32, device="cuda")
    last = B - 1
    results = {}
    for area, lvl in ((50.0, 4), (20.0, -1), (120.0, 2)):
        for name, kernel, mode in (("reordered", 0, 3), ("full sweep", 0, 1), ("popcount", 1, 1)):
            m.debug_set_sweep_kernel(kernel)
            m.debug_set_window_mode(mode)
            out.zero_(); nout.zero_()
            m.frame_match_chain_async(B, desc.data_ptr() + last * CAP * 32, kp.data_ptr() + last * CAP * KP, cnt.data_ptr() + 4 * last,
                                      desc.data_ptr(), kp.data_ptr(), CAP, cnt.data_ptr(), out, nout, 300.0, area, lvl, st)
            torch.cuda.synchronize()
            results[name] = (out.cpu().numpy().copy(), nout.cpu().numpy().copy())
        for name in ("full sweep", "popcount"):
            assert np.array_equal(results["reordered"][1], results[name][1]), (area, lvl, name)
            assert results["reordered"][0].tobytes() == results[name][0].tobytes(), (area, lvl, name)
        assert results["reordered"][1].min() > 200, (area, lvl)
        # frame 5 against frame 4 on the CPU
        k_all = kp.cpu().numpy().view(capi.KEYPOINT_DTYPE).reshape(B, CAP); d_all = desc.cpu().numpy().reshape(B, CAP, 32)
        ref = oracle.match_gated(d_all[4][:counts[4]], k_all[4][:counts[4]], d_all[5][:counts[5]], k_all[5][:counts[5]], 300.0, area, lvl)
        got = results["reordered"][0].view(capi.DMATCH_DTYPE)[5 * CAP: 5 * CAP + len(ref)]
        assert int(results["reordered"][1][5]) == len(ref) and got.tobytes() == ref.tobytes(), (area, lvl)
    m.close(); ex.close()


def test_windowed_batch_with_different_capacities(oracle, capi):
    """Batched problems whose two sets have different row pitches, window gate on: the reordered path against the oracle."""
    torch = _torch()
    st = torch.cuda.current_stream().cuda_stream
    B, CAP1, CAP2 = 3, 384, 640
    n1 = [384, 100, 257]; n2 = [640, 33, 500]
    sets = [_sets(CAP1, CAP2, 500 + b) for b in range(B)]
    D1 = torch.from_numpy(np.concatenate([s_[0] for s_ in sets])).cuda(); K1 = _dev_kp(torch, np.concatenate([s_[1] for s_ in sets]))
    D2 = torch.from_numpy(np.concatenate([s_[2] for s_ in sets])).cuda(); K2 = _dev_kp(torch, np.concatenate([s_[3] for s_ in sets]))
    N1 = torch.tensor(n1, dtype=torch.int32, device="cuda"); N2 = torch.tensor(n2, dtype=torch.int32, device="cuda")
    out = torch.zeros(B * CAP1 * DM, dtype=torch.uint8, device="cuda")
    nout = torch.zeros(B, dtype=torch.int32, device="cuda")
    m = capi.Matcher()
    for mode in (3, 2, 1):
        m.debug_set_window_mode(mode)
        m.frame_match_batch_async(B, D1, K1, CAP1, N1, D2, K2, CAP2, N2, out, nout, 300.0, 70.0, 3, st)
        torch.cuda.synchronize()
        res = out.cpu().numpy().view(capi.DMATCH_DTYPE)
        for b in range(B):
            d1, k1, d2, k2 = sets[b]
            ref = oracle.match_gated(d1[:n1[b]], k1[:n1[b]], d2[:n2[b]], k2[:n2[b]], 300.0, 70.0, 3)
            assert int(nout[b].item()) == len(ref), (mode, b)
            assert res[b * CAP1: b * CAP1 + len(ref)].tobytes() == ref.tobytes(), (mode, b)
    m.close()


def test_frame_match_async_and_batch(oracle, capi):
    torch = _torch()
    st = torch.cuda.current_stream().cuda_stream
    B, CAP = 4, 512
    m = capi.Matcher()
    m.reserve(B, CAP, CAP)
    n1 = [300, 512, 0, 77]; n2 = [410, 1, 100, 512]
    sets = [_sets(CAP, CAP, 50 + b) for b in range(B)]
    D1 = torch.from_numpy(np.concatenate([s[0] for s in sets])).cuda()
    K1 = _dev_kp(torch, np.concatenate([s[1] for s in sets]))
    D2 = torch.from_numpy(np.concatenate([s[2] for s in sets])).cuda()
    K2 = _dev_kp(torch, np.concatenate([s[3] for s in sets]))
    N1 = torch.tensor(n1, dtype=torch.int32, device="cuda"); N2 = torch.tensor(n2, dtype=torch.int32, device="cuda")
    out = torch.zeros(B * CAP * DM, dtype=torch.uint8, device="cuda")
    nout = torch.zeros(B, dtype=torch.int32, device="cuda")
    for area, lvl, mode in ((-1.0, -1, 0), (80.0, 4, 1), (80.0, 4, 2), (25.0, -1, 2), (80.0, 4, 3), (25.0, -1, 3), (80.0, 4, 0)):
        m.debug_set_window_mode(mode)
        m.frame_match_batch_async(B, D1, K1, CAP, N1, D2, K2, CAP, N2, out, nout, 300.0, area, lvl, st)
        torch.cuda.synchronize()
        res = out.cpu().numpy().view(capi.DMATCH_DTYPE)
        for b in range(B):
            d1, k1, d2, k2 = sets[b]
            ref = oracle.match_gated(d1[:n1[b]], k1[:n1[b]], d2[:n2[b]], k2[:n2[b]], 300.0, area, lvl)
            assert int(nout[b].item()) == len(ref)
            assert res[b * CAP: b * CAP + len(ref)].tobytes() == ref.tobytes(), (b, area, mode)
    m.debug_set_window_mode(2)
    # the two optional post-filters on the batched device path
    m.set_options(0.9, 30)
    m.reserve(B, CAP, CAP)
    m.frame_match_batch_async(B, D1, K1, CAP, N1, D2, K2, CAP, N2, out, nout, 300.0, 80.0, 4, st)
    torch.cuda.synchronize()
    res = out.cpu().numpy().view(capi.DMATCH_DTYPE)
    for b in range(B):
        d1, k1, d2, k2 = sets[b]
        ref = oracle.match_gated_ex(d1[:n1[b]], k1[:n1[b]], d2[:n2[b]], k2[:n2[b]], 300.0, 80.0, 4, 0.9, 30)
        assert int(nout[b].item()) == len(ref)
        assert res[b * CAP: b * CAP + len(ref)].tobytes() == ref.tobytes(), b
    with pytest.raises(capi.SlamAmdError):
        m.match_pairs_async(D1, K1, B, CAP, torch.zeros((1, 2), dtype=torch.int32, device="cuda"), 1,
                            torch.zeros((1, CAP), dtype=torch.int32, device="cuda"), 300.0, -1.0, -1, st)
    m.set_options(0.0, 0)
    m.debug_set_window_mode(0)
    # single problem, device counts smaller than the bounds, id gather
    d1, k1, d2, k2 = sets[0]
    ids = np.arange(5, 400, 7, dtype=np.int32)
    d_ids = torch.from_numpy(ids).cuda()
    m.frame_match_async(D1, K1, CAP, None, D2, K2, CAP, N2[:1], out, nout, d_ids=d_ids, n_ids=len(ids),
                        max_dist=300.0, area=-1.0, max_level=3, stream=st)
    torch.cuda.synchronize()
    ref = oracle.frame_match(d1, k1, d2[:n2[0]], k2[:n2[0]], ids, 300.0, -1.0, 3)
    assert int(nout[0].item()) == len(ref)
    assert out.cpu().numpy().view(capi.DMATCH_DTYPE)[:len(ref)].tobytes() == ref.tobytes()
    m.close()


def _pairs_reference(oracle, desc, kp, n_per, pairs, area, lvl, max_dist=300.0):
    out = np.full((len(pairs), n_per), -1, np.int64)
    for r, (i, j) in enumerate(pairs):
        m = oracle.match_gated(desc[i], kp[i], desc[j], kp[j], max_dist, area, lvl)
        out[r, m["queryIdx"]] = (m["distance"].astype(np.int64) << 16) | m["trainIdx"]
    return (out & 0xffffffff).astype(np.uint32)


def test_match_pairs_vs_oracle(oracle, capi):
    torch = _torch()
    from slam_amd import pairs as pairs_mod
    n_kf, n_per = 7, 530
    desc = synth.random_descriptors(n_kf * n_per, 3).reshape(n_kf, n_per, 32)
    desc[3, :100] = desc[1, 100:200]                      # some exact matches across keyframes
    kp = synth.random_keypoints(n_kf * n_per, seed=4).reshape(n_kf, n_per)
    plist = np.concatenate([pairs_mod.all_pairs(n_kf), [[4, 4], [6, 0]]]).astype(np.int32)
    m = capi.Matcher()
    d_desc = torch.from_numpy(desc.reshape(-1, 32)).cuda()
    d_kp = _dev_kp(torch, kp.reshape(-1))
    d_pairs = torch.from_numpy(plist).cuda()
    out = torch.zeros((len(plist), n_per), dtype=torch.int32, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    for area, lvl, md in ((-1.0, -1, 300.0), (-1.0, 4, 300.0), (150.0, 4, 300.0), (-1.0, -1, 100.0), (150.0, -1, 300.0),
                          (40.0, 4, 300.0), (40.0, 3, 60.0)):
        ref = _pairs_reference(oracle, desc, kp, n_per, plist.tolist(), area, lvl, md)
        # matrix cores (with a window: keyframes reordered by grid cell, then as they are), popcount
        for kernel, mode in ((0, 0), (0, 3), (0, 1), (1, 0)):
            m.debug_set_sweep_kernel(kernel)
            m.debug_set_window_mode(mode)
            out.zero_()
            m.match_pairs_async(d_desc, d_kp, n_kf, n_per, d_pairs, len(plist), out, md, area, lvl, st)
            torch.cuda.synchronize()
            got = out.cpu().numpy().view(np.uint32)
            assert np.array_equal(got, ref), (area, lvl, kernel, mode)
    m.close()


def test_match_pairs_full_size_properties(capi):
    """Multi-GPU benchmark configuration, size per pair (2000 x 2000); checked by recomputation on samples."""
    torch = _torch()
    from slam_amd import pairs as pairs_mod
    n_kf, n_per = 24, 2000
    desc = synth.random_descriptors(n_kf * n_per, 3).reshape(n_kf, n_per, 32)
    plist = pairs_mod.all_pairs(n_kf)
    m = capi.Matcher()
    d_desc = torch.from_numpy(desc.reshape(-1, 32)).cuda()
    out = torch.zeros((len(plist), n_per), dtype=torch.int32, device="cuda")
    m.match_pairs_async(d_desc, None, n_kf, n_per, torch.from_numpy(plist).cuda(), len(plist), out,
                        300.0, -1.0, -1, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    got = out.cpu().numpy().view(np.uint32)
    m.close()
    bits = np.unpackbits(desc, axis=2)                     # [kf, n_per, 256]
    rng = np.random.default_rng(0)
    for r in rng.integers(0, len(plist), 12):
        i, j = plist[r]
        dist = (bits[i][:, None, :16] != bits[j][None, :, :16]).sum(2)  # cheap partial check first
        full = np.zeros((n_per, n_per), np.int32)
        for c in range(0, 256, 64):
            full += (bits[i][:, None, c:c + 64] != bits[j][None, :, c:c + 64]).sum(2)
        best = full.min(1); arg = full.argmin(1)           # argmin = first (lowest index) minimum
        assert np.array_equal(got[r] >> 16, best.astype(np.uint32))
        assert np.array_equal(got[r] & 0xffff, arg.astype(np.uint32))


# ---------------------------------------------------- projection, duplicates ----
def test_project_points_bit_exact(oracle, capi):
    import math
    rng = np.random.default_rng(3)
    ang = 0.2
    R = np.array([[math.cos(ang), 0, math.sin(ang)], [0, 1, 0], [-math.sin(ang), 0, math.cos(ang)]])
    Rt = np.hstack([R, [[0.1], [-0.2], [0.5]]])
    K4 = [963.0, 950.4, 539.5, 404.5]
    X = (rng.random((20000, 3)) * [4, 3, 5] + [-2, -1.5, 2]).astype(np.float32)
    X[7] = [0.0, 0.0, -0.5 / math.cos(ang)]               # a point with camera-space z near 0
    m = capi.Matcher()
    for d5 in (None, [0.1, -0.05, 0.001, 0.002, 0.01]):
        got = m.project_points(Rt, K4, d5, X)
        ref = oracle.project_points(Rt, K4, d5, X)
        assert got.tobytes() == ref.tobytes()
    assert len(m.project_points(Rt, K4, None, X[:0])) == 0
    m.close()


def test_duplicate_pairs_vs_oracle(oracle, capi):
    d1 = synth.random_descriptors(300, 1)
    d2 = np.concatenate([synth.noisy_copy(d1[:100], 30, 2), synth.random_descriptors(231, 3)])
    rng = np.random.default_rng(4)
    p1 = rng.random((300, 3)).astype(np.float32)
    p2 = rng.random((331, 3)).astype(np.float32)
    p2[:60] = p1[:60] + rng.normal(0, 4e-3, (60, 3)).astype(np.float32)
    m = capi.Matcher()
    got = m.duplicate_pairs(d1, p1, d2, p2, 100, 1e-2)
    m.close()
    ref = oracle.duplicate_pairs(d1, p1, d2, p2, 100, 1e-2)
    assert np.array_equal(got, ref) and ref.sum() > 10


# ------------------------------------------------------------------ pipeline ----
@pytest.mark.parametrize("batch", [1, 3])
def test_frame_pipeline_matches_sequential_oracle(oracle, capi, batch):
    """The library's three-stream pipeline (H2D / extract + match / D2H) over a short sequence, frames queued one
    at a time and three at a time (7 frames: the last batch is incomplete and flushed)."""
    from slam_amd.stream import FramePipeline
    W, H, NF = 320, 240, 400
    frames = synth.frame_sequence(W, H, 7, seed=5)
    pipe = FramePipeline(W, H, capi.orb_params(nfeatures=NF), 2048, depth=3, batch=batch,
                         max_dist=300.0, area=50.0, max_level=4)
    if batch > 1:
        pipe.push(frames[0])
        with pytest.raises(capi.SlamAmdError):      # its batch is neither complete nor flushed
            pipe.result(0)
        pipe.close()
        pipe = FramePipeline(W, H, capi.orb_params(nfeatures=NF), 2048, depth=3, batch=batch,
                             max_dist=300.0, area=50.0, max_level=4)
    got = pipe.run(frames, keep_results=True)
    pipe.close()
    po = oracle.params(nfeatures=NF)
    prev = None
    assert len(got) == len(frames)
    for i, f in enumerate(frames):
        kp_o, d_o = oracle.detect_and_compute(po, f)
        kp, d, mt = got[i]
        assert kp.tobytes() == kp_o.tobytes() and np.array_equal(d, d_o), i
        if prev is not None:
            ref = oracle.match_gated(prev[1], prev[0], d_o, kp_o, 300.0, 50.0, 4)
            assert mt.tobytes() == ref.tobytes(), i
        prev = (kp_o, d_o)


def test_frame_pipeline_slots_of_more_than_64_frames(oracle, capi):
    """Slots of 72 frames (the per-slot counts used to be a fixed 64-entry array): 150 frames of a short sequence
    repeated, two full slots and a flushed one; every frame's results equal those of the one-frame-per-slot pipeline,
    and three of them (first of a slot, last of a slot, in the flushed slot) equal the oracle's."""
    from slam_amd.stream import FramePipeline
    W, H, NF, N = 160, 120, 150, 150
    base = synth.frame_sequence(W, H, 6, seed=9)
    frames = [base[i % 6] for i in range(N)]
    res = {}
    for batch in (1, 72):
        pipe = FramePipeline(W, H, capi.orb_params(nfeatures=NF), 512, depth=3, batch=batch, max_dist=300.0, area=-1.0, max_level=4)
        res[batch] = pipe.run(frames, keep_results=True)
        pipe.close()
    assert len(res[1]) == N and len(res[72]) == N
    for i in range(N):
        for a, b in zip(res[1][i], res[72][i]):
            assert a.tobytes() == b.tobytes(), i
    po = oracle.params(nfeatures=NF)
    for i in (72, 143, 149):
        kp_p, d_p = oracle.detect_and_compute(po, frames[i - 1])
        kp_o, d_o = oracle.detect_and_compute(po, frames[i])
        kp, d, mt = res[72][i]
        assert kp.tobytes() == kp_o.tobytes() and np.array_equal(d, d_o), i
        assert mt.tobytes() == oracle.match_gated(d_p, kp_p, d_o, kp_o, 300.0, -1.0, 4).tobytes(), i

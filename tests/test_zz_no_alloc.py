"""The entry point that allocates nothing: slam_amd_frame_match_noalloc_async.

Its scratch is a fixed array inside the library instead of buffers the matcher allocates;
its results must equal those of slam_amd_frame_match_async on the same inputs, byte for
byte, also on a second call with a smaller input (the array keeps the first call's
contents).  Collected last, so that a failure here cannot hide another test under -x."""
import numpy as np
import pytest

from slam_amd import synth

pytestmark = pytest.mark.gpu
DM = 16


def _dev_kp(t, arr):
    return t.from_numpy(np.ascontiguousarray(arr).view(np.uint8).copy()).cuda()


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


def _both(torch, capi, m, d1, k1, d2, k2, nq, nt, ids, area, lvl, ratio_on):
    """Runs the allocating and the non-allocating entry point on the first nq x nt rows."""
    st = torch.cuda.current_stream().cuda_stream
    D1 = torch.from_numpy(d1[:nq].copy()).cuda(); K1 = _dev_kp(torch, k1[:nq])
    D2 = torch.from_numpy(d2[:nt].copy()).cuda(); K2 = _dev_kp(torch, k2[:nt])
    d_ids = None if ids is None else torch.from_numpy(ids).cuda()
    n_ids = 0 if ids is None else len(ids)
    rows = max(nq, n_ids, 1)
    res = []
    for fn in (m.frame_match_async, m.frame_match_noalloc_async):
        out = torch.zeros(rows * DM, dtype=torch.uint8, device="cuda")
        nout = torch.full((1,), -7, dtype=torch.int32, device="cuda")
        torch.cuda.synchronize()
        free0 = torch.cuda.mem_get_info()[0]
        fn(D1, K1, nq, None, D2, K2, nt, None, out, nout, d_ids=d_ids, n_ids=n_ids,
           max_dist=300.0, area=area, max_level=lvl, stream=st)
        torch.cuda.synchronize()
        free1 = torch.cuda.mem_get_info()[0]
        n = int(nout.item())
        res.append((n, out.cpu().numpy().view(capi.DMATCH_DTYPE)[:max(n, 0)].copy(), free0 - free1))
    return res


def test_noalloc_entry_point_equals_frame_match_async(oracle, capi):
    import torch
    m = capi.Matcher()
    d1, k1, d2, k2 = _sets(5000, 3000, 77)
    ids_big = np.arange(3, 5000, 2, dtype=np.int32)
    ids_small = np.arange(0, 700, 3, dtype=np.int32)
    # two calls in a row per configuration, the second with a smaller input
    for (area, lvl, ratio, histo) in ((-1.0, -1, 0.0, 0), (60.0, 4, 0.0, 0), (60.0, 3, 0.85, 30)):
        m.set_options(ratio, histo)
        m.reserve(1, 5000, 3000)            # the allocating call's scratch; the other call has none to size
        for (nq, nt, ids) in ((5000, 3000, None), (777, 129, None), (5000, 3000, ids_big), (700, 64, ids_small)):
            (n_a, got_a, _), (n_b, got_b, grew) = _both(torch, capi, m, d1, k1, d2, k2, nq, nt, ids, area, lvl, ratio > 0)
            assert n_a == n_b and n_a > 0, (area, lvl, ratio, nq, nt)
            assert got_a.tobytes() == got_b.tobytes(), (area, lvl, ratio, nq, nt)
            assert grew == 0, "device memory in use changed across the non-allocating call"
            # and both equal the oracle
            q = slice(None) if ids is None else ids
            if ratio > 0:
                ref = oracle.match_gated_ex(d1[:nq][q], k1[:nq][q], d2[:nt], k2[:nt], 300.0, area, lvl, ratio, histo)
                if ids is not None:
                    ref["queryIdx"] = ids[ref["queryIdx"]]
            else:
                ref = oracle.frame_match(d1[:nq], k1[:nq], d2[:nt], k2[:nt], ids, 300.0, area, lvl)
            assert got_b.tobytes() == ref.tobytes(), (area, lvl, ratio, nq, nt)
    m.set_options(0.0, 0)
    # beyond the fixed maximum: a clear error, nothing launched
    st = torch.cuda.current_stream().cuda_stream
    dummy = torch.zeros(64, dtype=torch.uint8, device="cuda")
    with pytest.raises(capi.SlamAmdError) as e:
        m.frame_match_noalloc_async(dummy, None, 32769, None, dummy, None, 1, None, dummy, dummy,
                                    max_dist=300.0, area=-1.0, max_level=-1, stream=st)
    assert e.value.status == capi.ERR_CAPACITY
    m.close()

#!/usr/bin/env python3
"""Times the matching kernels alone (HIP events via the library's stage timer) on the shapes of the
benchmark configurations: 64 frames x ~2000 keypoints chained (every gate combination the reference's
call sites use), 20000 x 4000 with projected positions, 10000 x 10000, and keyframe pairs.
Usage: python tools/bench_match_frames.py [kernel]   (kernel 0 = matrix cores, 1 = popcount)"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from slam_amd import capi, pairs as pairs_mod, synth

kernel = int(sys.argv[1]) if len(sys.argv) > 1 else 0
only_frames = len(sys.argv) > 2 and sys.argv[2] == "frames"
dev = torch.device("cuda", 0)
st = torch.cuda.current_stream().cuda_stream
KP, DM = 28, 16


warm_us = 30000.0


def timed(label, fn, cmp_count, stages=("match", "spatial_sort", "match_window", "match_pairs"), reps=20):
    for _ in range(max(3, int(warm_us / max(cmp_count / 3e6, 20.0)))):   # keep the chip busy long enough to clock up
        fn()
    torch.cuda.synchronize()
    capi.prof_enable(8 * reps + 16)
    torch.cuda._sleep(int(2.0e6) * (1 + reps // 8))   # the launches below queue up behind this: events bracket kernels, not host gaps
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    res = capi.prof_collect()
    capi.prof_enable(0)
    us = {k: res[k][0] / res[k][1] * 1e3 for k in res if k in stages}
    total = sum(us.values())
    print("%-44s %8.1f us  %7.2f Tcmp/s   %s" % (label, total, cmp_count / total / 1e6,
                                                " ".join("%s=%.1f" % kv for kv in us.items())), flush=True)
    return total


# ---- 64 frames of 640x480, as the extractor leaves them (sorted by octave, real positions) ----
W, H, B, CAP = 640, 480, int(os.environ.get("SA_BENCH_BATCH", "64")), 4096
ex = capi.Extractor(capi.orb_params(nfeatures=2000), W, H, max_batch=B, max_keypoints=CAP)
frames = synth.frame_sequence(W, H, B, seed=1)
d_img = torch.from_numpy(np.stack(frames)).to(dev)
kp = torch.zeros(B * CAP * KP, dtype=torch.uint8, device=dev)
desc = torch.zeros(B * CAP * 32, dtype=torch.uint8, device=dev)
cnt = torch.zeros(B, dtype=torch.int32, device=dev)
ex.extract_batch_async(d_img, B, H * W, W, kp.data_ptr(), desc.data_ptr(), cnt.data_ptr(), CAP, st)
torch.cuda.synchronize()
counts = cnt.cpu().numpy().astype(np.float64)
cmp_frames = float((counts * np.roll(counts, 1)).sum())
out = torch.zeros(B * CAP * DM, dtype=torch.uint8, device=dev)
nout = torch.zeros(B, dtype=torch.int32, device=dev)
m = capi.Matcher()
m.debug_set_sweep_kernel(kernel)
m.reserve(B, CAP, CAP)
last = B - 1
for area, lvl, mode in ((-1.0, -1, 0), (-1.0, 4, 0), (-1.0, 3, 0), (50.0, 4, 1), (50.0, 4, 3), (50.0, -1, 1), (50.0, -1, 3)):
    m.debug_set_window_mode(mode)      # 1: the sets as they are, 3: reordered by grid cell first
    def chain():
        m.frame_match_chain_async(B, desc.data_ptr() + last * CAP * 32, kp.data_ptr() + last * CAP * KP,
                                  cnt.data_ptr() + 4 * last, desc.data_ptr(), kp.data_ptr(), CAP, cnt.data_ptr(),
                                  out, nout, max_dist=300.0, area=area, max_level=lvl, stream=st)
    timed("%d frames chained  area=%g max_level=%d%s" % (B, area, lvl, " reordered" if mode == 3 else ""), chain, cmp_frames)
    print("    matches per frame: %.1f" % float(nout.cpu().numpy().mean()))
mr = capi.Matcher()
mr.debug_set_sweep_kernel(kernel)
mr.set_options(0.8, 0)
mr.reserve(B, CAP, CAP)
for area, lvl in ((-1.0, -1), (-1.0, 4)):
    def chain_r():
        mr.frame_match_chain_async(B, desc.data_ptr() + last * CAP * 32, kp.data_ptr() + last * CAP * KP,
                                   cnt.data_ptr() + 4 * last, desc.data_ptr(), kp.data_ptr(), CAP, cnt.data_ptr(),
                                   out, nout, max_dist=300.0, area=area, max_level=lvl, stream=st)
    timed("%d frames chained ratio 0.8  area=%g lvl=%d" % (B, area, lvl), chain_r, cmp_frames)
m.debug_set_window_mode(0)
ex.close(); mr.close()
if only_frames:
    m.close()
    sys.exit(0)

# ---- one large problem: 20000 x 4000 with a window (projection), 10000 x 10000 ----
for nq, nt, w, h in ((20000, 4000, 1241, 376), (10000, 10000, 3840, 2160)):
    d2 = synth.random_descriptors(nt, 7)
    rng = np.random.default_rng(8)
    src = rng.integers(0, nt, size=nq)
    d1 = synth.noisy_copy(d2[src], 25, 9)
    k1 = synth.random_keypoints(nq, w, h, seed=10)
    k2 = synth.random_keypoints(nt, w, h, seed=11)
    k2 = k2[np.lexsort((k2["x"], k2["y"], k2["octave"]))]          # as an extractor orders them
    k1["x"] = k2["x"][src] + rng.normal(0, 20, nq).astype(np.float32)
    k1["y"] = k2["y"][src] + rng.normal(0, 20, nq).astype(np.float32)
    D1 = torch.from_numpy(d1).to(dev); D2 = torch.from_numpy(d2).to(dev)
    K1 = torch.from_numpy(k1.view(np.uint8).copy()).to(dev); K2 = torch.from_numpy(k2.view(np.uint8).copy()).to(dev)
    o = torch.zeros(nq * DM, dtype=torch.uint8, device=dev)
    n = torch.zeros(1, dtype=torch.int32, device=dev)
    m.reserve(1, nq, nt)
    for area, lvl, mode in ((-1.0, -1, 1), (-1.0, 4, 1), (50.0, 4, 1), (50.0, 4, 2), (50.0, 4, 3)):
        m.debug_set_window_mode(mode)
        def one():
            m.frame_match_async(D1, K1, nq, None, D2, K2, nt, None, o, n, max_dist=300.0, area=area, max_level=lvl, stream=st)
        timed("%d x %d area=%g lvl=%d %s" % (nq, nt, area, lvl, {1: "full sweep", 2: "cell-sorted, popcount", 3: "cell-sorted, matrix cores"}[mode]), one, float(nq) * nt)
m.debug_set_window_mode(0)

# ---- keyframe pairs ----
n_kf, n_per = 96, 2000
pdesc = torch.from_numpy(synth.random_descriptors(n_kf * n_per, 3)).to(dev)
pkp = torch.from_numpy(synth.random_keypoints(n_kf * n_per, seed=4).view(np.uint8)).to(dev)
plist = pairs_mod.all_pairs(n_kf)
d_pairs = torch.from_numpy(plist).to(dev)
pout = torch.empty((len(plist), n_per), dtype=torch.int32, device=dev)
for area, lvl in ((-1.0, -1), (-1.0, 4), (50.0, 4)):
    def prs():
        m.match_pairs_async(pdesc, pkp, n_kf, n_per, d_pairs, len(plist), pout, 300.0, area, lvl, st)
    timed("%d pairs x %d^2 area=%g lvl=%d" % (len(plist), n_per, area, lvl), prs, len(plist) * float(n_per) * n_per, reps=3)
m.close()

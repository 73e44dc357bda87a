#!/usr/bin/env python3
"""Diagnostic: wave timelines of the matrix-core matcher from a library built with -DSA_MM_STAMPS
(SLAM_AMD_LIB=build_variants/lib_stamps.so python tools/mm_stamps.py)."""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from slam_amd import capi, synth

dev = torch.device("cuda", 0)
st = torch.cuda.current_stream().cuda_stream
KP, DM = 28, 16
W, H, B, CAP = 640, 480, 64, 4096
ex = capi.Extractor(capi.orb_params(nfeatures=2000), W, H, max_batch=B, max_keypoints=CAP)
frames = synth.frame_sequence(W, H, 8, seed=1)
d_img = torch.from_numpy(np.stack([frames[i % 8] for i in range(B)])).to(dev)
kp = torch.zeros(B * CAP * KP, dtype=torch.uint8, device=dev)
desc = torch.zeros(B * CAP * 32, dtype=torch.uint8, device=dev)
cnt = torch.zeros(B, dtype=torch.int32, device=dev)
out = torch.zeros(B * CAP * DM, dtype=torch.uint8, device=dev)
nout = torch.zeros(B, dtype=torch.int32, device=dev)
m = capi.Matcher()
m.reserve(B, CAP, CAP)
last = B - 1
for it in range(6):     # like the benchmark step: extraction, then the chained match
    ex.extract_batch_async(d_img, B, H * W, W, kp.data_ptr(), desc.data_ptr(), cnt.data_ptr(), CAP, st)
    m.frame_match_chain_async(B, desc.data_ptr() + last * CAP * 32, kp.data_ptr() + last * CAP * KP,
                              cnt.data_ptr() + 4 * last, desc.data_ptr(), kp.data_ptr(), CAP, cnt.data_ptr(),
                              out, nout, max_dist=300.0, area=float(os.environ.get("SA_STAMP_AREA", "-1")), max_level=int(os.environ.get("SA_STAMP_LEVEL", "-1")), stream=st)
torch.cuda.synchronize()
n = 2048
buf = np.zeros((n, 10), np.uint64)
lib = capi.lib()
lib.slam_amd_debug_mm_stamps.argtypes = [C.c_void_p, C.c_int]
assert lib.slam_amd_debug_mm_stamps(buf.ctypes.data, n) == 0
t = buf[:, :4].astype(np.int64)
rt = buf[:, 5].astype(np.int64)
hw = buf[:, 4].astype(np.int64)
wave_id, simd, cu, sh, se = hw & 15, (hw >> 4) & 3, (hw >> 8) & 15, (hw >> 12) & 1, (hw >> 13) & 7
print("waves", n, "keypoints", cnt.cpu().numpy()[:4])
pro, loop, tail = t[:, 1] - t[:, 0], t[:, 2] - t[:, 1], t[:, 3] - t[:, 2]
for name, v in (("prologue", pro), ("tile loop", loop), ("last tile + output", tail), ("whole wave", t[:, 3] - t[:, 0])):
    print("%-20s cycles: min %7d  median %7d  p90 %7d  max %7d" % (name, v.min(), np.median(v), np.percentile(v, 90), v.max()))
pt = buf[:, 6:10].astype(np.int64)
for name, v in (("  loads issued", pt[:, 0] - t[:, 0]), ("  counts arrived", pt[:, 1] - pt[:, 0]), ("  queries expanded", pt[:, 2] - pt[:, 1]),
                ("  tiles staged + barrier", pt[:, 3] - pt[:, 2]), ("  operands read", t[:, 1] - pt[:, 3])):
    print("%-26s cycles: min %7d  median %7d  p90 %7d  max %7d" % (name, v.min(), np.median(v), np.percentile(v, 90), v.max()))
print("real-time clock at wave start (10 ns ticks): spread %d" % (rt.max() - rt.min()))
# shader clocks are per XCD: compare within a wave only.  Start skew in real time:
order = np.argsort(rt)
print("start real-time percentiles (us after the first): p50 %.2f p90 %.2f max %.2f" % tuple(
    (np.percentile(rt, q) - rt.min()) / 100.0 for q in (50, 90, 100)))
tiles = np.ceil(cnt.cpu().numpy().astype(np.float64) / 128).mean()
print("tiles per wave ~%.1f -> cycles per tile in the loop: median %.0f (products alone: 1024 for 128 rows)" % (tiles, np.median(loop) / max(tiles - 1, 1)))
# co-residency: how many waves share (se, sh, cu, simd)
key = ((se * 2 + sh) * 16 + cu) * 4 + simd
uniq, c = np.unique(key, return_counts=True)
print("distinct SIMD ids seen: %d (of the XCD-local id space), waves per id: min %d max %d" % (len(uniq), c.min(), c.max()))

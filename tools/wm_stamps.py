#!/usr/bin/env python3
"""Diagnostic: wave timelines of the reordered window matcher (library built with -DSA_WM_STAMPS:
tools/build_variant.sh wmstamps slam_amd/csrc/match_mfma_kernels.hip -DSA_WM_STAMPS, run with
SLAM_AMD_LIB=build_variants/lib_wmstamps.so).  64 frames of 640 x 480, chained, window 50."""
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
W, H, B, CAP = 640, 480, int(os.environ.get("SA_BENCH_BATCH", "64")), 4096
ex = capi.Extractor(capi.orb_params(nfeatures=2000), W, H, max_batch=B, max_keypoints=CAP)
d_img = torch.from_numpy(np.stack(synth.frame_sequence(W, H, B, seed=1))).to(dev)
kp = torch.zeros(B * CAP * KP, dtype=torch.uint8, device=dev)
desc = torch.zeros(B * CAP * 32, dtype=torch.uint8, device=dev)
cnt = torch.zeros(B, dtype=torch.int32, device=dev)
ex.extract_batch_async(d_img, B, H * W, W, kp.data_ptr(), desc.data_ptr(), cnt.data_ptr(), CAP, st)
torch.cuda.synchronize()
out = torch.zeros(B * CAP * DM, dtype=torch.uint8, device=dev)
nout = torch.zeros(B, dtype=torch.int32, device=dev)
m = capi.Matcher()
m.reserve(B, CAP, CAP)
m.debug_set_window_mode(3)
lib = capi.lib()
NW = 8192
lib.slam_amd_debug_wm_stamps.argtypes = [C.c_void_p, C.c_int]
buf = np.zeros((NW, 6), dtype=np.uint64)
last = B - 1
area = float(os.environ.get("SA_STAMP_AREA", "50"))
lvl = int(os.environ.get("SA_STAMP_LEVEL", "-1"))
for it in range(30):
    m.frame_match_chain_async(B, desc.data_ptr() + last * CAP * 32, kp.data_ptr() + last * CAP * KP, cnt.data_ptr() + 4 * last,
                              desc.data_ptr(), kp.data_ptr(), CAP, cnt.data_ptr(), out, nout, max_dist=300.0, area=area,
                              max_level=lvl, stream=st)
torch.cuda.synchronize()
lib.slam_amd_debug_wm_stamps(buf.ctypes.data, NW)
s = buf.astype(np.int64)
ok = s[:, 2] > s[:, 0]
s = s[ok]
t0 = s[:, 0].min()
life = s[:, 2] - s[:, 0]
loop = s[:, 2] - s[:, 1]
nb = np.maximum(s[:, 3], 1)
print("waves %d; kernel span %d cycles (first start to last end)" % (len(s), s[:, 2].max() - t0))
print("start spread: median %d max %d" % (np.median(s[:, 0] - t0), (s[:, 0] - t0).max()))
print("prologue: median %d  max %d" % (np.median(s[:, 1] - s[:, 0]), (s[:, 1] - s[:, 0]).max()))
print("loop: median %d max %d; blocks median %d max %d; cycles per block median %d" %
      (np.median(loop), loop.max(), np.median(s[:, 3]), s[:, 3].max(), np.median(loop / nb)))
print("waiting for fetches at the loop top: median %d per block (%.0f %% of the loop)" %
      (np.median(s[:, 4] / nb), 100.0 * s[:, 4].sum() / loop.sum()))
simd = s[:, 5] & 0xfffff      # wave slot, SIMD, CU, SH, SE (not XCC)
print("end time percentiles (cycles after first start): 50 %% %d, 90 %% %d, 99 %% %d, 100 %% %d" %
      tuple(np.percentile(s[:, 2] - t0, [50, 90, 99, 100])))
m.close(); ex.close()

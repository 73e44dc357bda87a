#!/usr/bin/env python3
"""Diagnostic: how many train blocks the waves of the reordered window matcher work on (library built with
-DSA_WM_COUNT: tools/build_variant.sh wmcount slam_amd/csrc/match_mfma_kernels.hip -DSA_WM_COUNT, run with
SLAM_AMD_LIB=build_variants/lib_wmcount.so).  64 frames of 640 x 480, chained, window 50."""
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
lib.slam_amd_debug_wm_count.argtypes = [C.POINTER(C.c_ulonglong), C.c_int]
buf = (C.c_ulonglong * 4)()
last = B - 1
for area in (50.0, 25.0, 100.0):
    lib.slam_amd_debug_wm_count(buf, 1)
    m.frame_match_chain_async(B, desc.data_ptr() + last * CAP * 32, kp.data_ptr() + last * CAP * KP, cnt.data_ptr() + 4 * last,
                              desc.data_ptr(), kp.data_ptr(), CAP, cnt.data_ptr(), out, nout, max_dist=300.0, area=area,
                              max_level=-1, stream=st)
    torch.cuda.synchronize()
    lib.slam_amd_debug_wm_count(buf, 0)
    waves, liveb, allb, mx = [int(v) for v in buf]
    print("window %5.1f: %d waves, %.1f of %.1f blocks per wave (%.1f %%), longest list %d" %
          (area, waves, liveb / waves, allb / waves, 100.0 * liveb / allb, mx))
m.close(); ex.close()

#!/usr/bin/env python3
"""Consistency run of the streaming pipeline with the window gate on (both sets reordered by grid cell per batch):
320 frames of 640 x 480 through slots of 16 frames and of one frame give the same bytes, a rerun gives the same
bytes, and the matches of a frame lie inside the window.  Usage (GPU box): python tools/soak_window_pipeline.py"""
import sys, os, numpy as np
sys.path.insert(0, os.getcwd())
from slam_amd import capi, synth
from slam_amd.stream import FramePipeline
W, H, NF, N = 640, 480, 2000, 320
base = synth.frame_sequence(W, H, 40, seed=33)
frames = [base[i % 40] for i in range(N)]
res = {}
for batch, area in ((16, 50.0), (1, 50.0), (16, -1.0)):
    pipe = FramePipeline(W, H, capi.orb_params(nfeatures=NF), 4096, depth=3, batch=batch, max_dist=300.0, area=area, max_level=4)
    res[(batch, area)] = pipe.run(frames, keep_results=True)
    pipe.close()
a, b = res[(16, 50.0)], res[(1, 50.0)]
bad = 0
for i in range(N):
    for x, y in zip(a[i], b[i]):
        if x.tobytes() != y.tobytes(): bad += 1
print("frames", N, "mismatching arrays between batch 16 and batch 1 (window 50):", bad)
# window results are a subset-consistent: every windowed match must also satisfy the window
import numpy as np
m = a[100][2]; kp_prev = a[99][0]; kp = a[100][0]
dx = kp_prev["x"][m["queryIdx"]] - kp["x"][m["trainIdx"]]; dy = kp_prev["y"][m["queryIdx"]] - kp["y"][m["trainIdx"]]
print("matches", len(m), "max |dx|,|dy|", float(np.abs(dx).max()), float(np.abs(dy).max()))
# repeat the windowed batch-16 run: bitwise reproducible
pipe = FramePipeline(W, H, capi.orb_params(nfeatures=NF), 4096, depth=3, batch=16, max_dist=300.0, area=50.0, max_level=4)
c = pipe.run(frames, keep_results=True); pipe.close()
print("rerun differs:", sum(x.tobytes() != y.tobytes() for i in range(N) for x, y in zip(a[i], c[i])))

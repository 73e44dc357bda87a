#!/usr/bin/env python3
"""Times the keyframe-pair matching kernel alone (HIP events via the library's stage timer)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from slam_amd import capi, pairs as pairs_mod, synth

n_kf = int(sys.argv[1]) if len(sys.argv) > 1 else 128
n_per = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
dev = torch.device("cuda", 0)
m = capi.Matcher(0)
desc = torch.from_numpy(synth.random_descriptors(n_kf * n_per, 3)).to(dev)
kp = torch.from_numpy(synth.random_keypoints(n_kf * n_per, seed=4).view(np.uint8)).to(dev)
pairs = pairs_mod.all_pairs(n_kf)
d_pairs = torch.from_numpy(pairs).to(dev)
out = torch.empty((len(pairs), n_per), dtype=torch.int32, device=dev)
st = torch.cuda.current_stream().cuda_stream
for area, lvl in ((-1.0, -1), (-1.0, 4), (50.0, 4)):
    m.match_pairs_async(desc, kp, n_kf, n_per, d_pairs, len(pairs), out, 300.0, area, lvl, st)
    torch.cuda.synchronize()
    capi.prof_enable(16)
    for _ in range(3):
        m.match_pairs_async(desc, kp, n_kf, n_per, d_pairs, len(pairs), out, 300.0, area, lvl, st)
    torch.cuda.synchronize()
    ms, calls = capi.prof_collect()["match_pairs"]
    capi.prof_enable(0)
    cmp_ = len(pairs) * float(n_per) * n_per
    print("pairs=%d n_per=%d area=%g max_level=%d: %.3f ms/launch  %.1f Gcmp/s" %
          (len(pairs), n_per, area, lvl, ms / calls, cmp_ / (ms / calls) / 1e6))

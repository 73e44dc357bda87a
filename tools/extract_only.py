"""Runs the extraction of the benchmark's 64 frames 30 times (for profiling kernel variants)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from slam_amd import capi, synth
W, H, B, CAP = 640, 480, int(os.environ.get("SA_BENCH_BATCH", "64")), 4096
ex = capi.Extractor(capi.orb_params(nfeatures=2000), W, H, max_batch=B, max_keypoints=CAP)
d_img = torch.from_numpy(np.stack(synth.frame_sequence(W, H, B, seed=1))).cuda()
kp = torch.zeros(B * CAP * 28, dtype=torch.uint8, device="cuda"); desc = torch.zeros(B * CAP * 32, dtype=torch.uint8, device="cuda")
cnt = torch.zeros(B, dtype=torch.int32, device="cuda")
st = torch.cuda.current_stream().cuda_stream
for _ in range(30):
    ex.extract_batch_async(d_img, B, H * W, W, kp.data_ptr(), desc.data_ptr(), cnt.data_ptr(), CAP, st)
torch.cuda.synchronize()
print("keypoints per frame %.1f" % float(cnt.float().mean()))

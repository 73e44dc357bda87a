"""Runs one batch of the benchmark frames through a diagnostic build of the library whose fast_nms
kernel prints the shader clock at its phase boundaries for a few tiles (columns: staging, barrier,
compass test + compaction, barrier, ring test, barrier, score + vertical blur, barrier, suppression,
horizontal blur, tail).

Build (not shipped):  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -DSA_FAST_EXP=1 \
                            -c slam_amd/csrc/extract_kernels.hip -o /tmp/ek.o   and link it in place of extract_kernels.o
Run on the GPU box:   SLAM_AMD_LIB=<that library> python tools/fast_phase_stamps.py"""
import sys, numpy as np, torch
sys.path.insert(0, '.')
from slam_amd import synth, capi
# optional arguments: width height nfeatures batch capacity
W, H, NF, B, CAP = (int(v) for v in (sys.argv[1:6] + ['640', '480', '2000', '64', '4096'][len(sys.argv) - 1:]))
seq = synth.frame_sequence(W, H, 8, seed=1)
frames = np.stack([seq[i % 8] for i in range(B)])
d_img = torch.from_numpy(frames).cuda()
ex = capi.Extractor(capi.orb_params(nfeatures=NF), W, H, max_batch=B, max_keypoints=CAP)
d_kp = torch.zeros(B * CAP * 28, dtype=torch.uint8, device='cuda'); d_desc = torch.zeros(B * CAP * 32, dtype=torch.uint8, device='cuda')
d_cnt = torch.zeros(B, dtype=torch.int32, device='cuda')
for _ in range(2):
    ex.extract_batch_async(d_img, B, W * H, W, d_kp.data_ptr(), d_desc.data_ptr(), d_cnt.data_ptr(), CAP, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()

"""Timing experiment: the bench step (extract a batch, match every frame with the one before) split
into `ways` contiguous sub-batches that run on streams of their own, against the step as one batch.

    python tools/overlap_experiment.py [--ways 1 2 4] [--steps 20]

Every sub-batch has an extractor and a matcher of its own; sub-batch k's matching waits for the
extraction of sub-batch k - 1 (its first frame is matched with that sub-batch's last frame).
Prints ms per step for each split, eager and as a HIP graph replay.
"""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ways", type=int, nargs="+", default=[1, 2, 4])
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--match-joined", action="store_true", help="one matching call after all extractions")
    args = ap.parse_args()
    import torch
    from slam_amd import capi, synth

    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    W, H, NF, B, CAP = 640, 480, 2000, args.batch, 4096
    KP, DM = 28, 16
    seq = synth.frame_sequence(W, H, 8, seed=1)
    frames = np.stack([seq[i % 8] for i in range(B)])
    d_img = torch.from_numpy(frames).to(dev)
    d_kp = torch.zeros((B + 1) * CAP * KP, dtype=torch.uint8, device=dev)
    d_desc = torch.zeros((B + 1) * CAP * 32, dtype=torch.uint8, device=dev)
    d_cnt = torch.zeros(B + 1, dtype=torch.int32, device=dev)
    d_out = torch.zeros(B * CAP * DM, dtype=torch.uint8, device=dev)
    d_nout = torch.zeros(B, dtype=torch.int32, device=dev)
    kp_p, desc_p, cnt_p = d_kp.data_ptr(), d_desc.data_ptr(), d_cnt.data_ptr()
    img_p, out_p, nout_p = d_img.data_ptr(), d_out.data_ptr(), d_nout.data_ptr()
    params = capi.orb_params(nfeatures=NF)
    ref = None
    for ways in args.ways:
        nb = B // ways
        exs = [capi.Extractor(params, W, H, max_batch=nb, max_keypoints=CAP, device=0) for _ in range(ways)]
        ms = [capi.Matcher(device=0) for _ in range(ways)]
        for m in ms:
            m.reserve(B if args.match_joined else nb, CAP, CAP)
        side = [torch.cuda.Stream(device=dev) for _ in range(ways - 1)]

        def step():
            main_s = torch.cuda.current_stream()
            streams = [main_s] + side
            for s in side:
                s.wait_stream(main_s)
            ev = []
            for k in range(ways):
                b0 = k * nb
                with torch.cuda.stream(streams[k]):
                    st = streams[k].cuda_stream
                    exs[k].extract_batch_async(img_p + b0 * H * W, nb, H * W, W, kp_p + (b0 + 1) * CAP * KP,
                                               desc_p + (b0 + 1) * CAP * 32, cnt_p + 4 * (b0 + 1), CAP, st)
                    e = torch.cuda.Event()
                    e.record(streams[k])
                    ev.append(e)
                    if not args.match_joined:
                        if k > 0:
                            streams[k].wait_event(ev[k - 1])
                        ms[k].frame_match_batch_async(nb, desc_p + b0 * CAP * 32, kp_p + b0 * CAP * KP, CAP,
                                                      cnt_p + 4 * b0, desc_p + (b0 + 1) * CAP * 32,
                                                      kp_p + (b0 + 1) * CAP * KP, CAP, cnt_p + 4 * (b0 + 1),
                                                      out_p + b0 * CAP * DM, nout_p + 4 * b0, max_dist=300.0,
                                                      area=-1.0, max_level=-1, stream=st)
            for s in side:
                main_s.wait_stream(s)
            if args.match_joined:
                ms[0].frame_match_batch_async(B, desc_p, kp_p, CAP, cnt_p, desc_p + CAP * 32, kp_p + CAP * KP, CAP,
                                              cnt_p + 4, out_p, nout_p, max_dist=300.0, area=-1.0, max_level=-1,
                                              stream=main_s.cuda_stream)
            d_kp[:CAP * KP].copy_(d_kp[B * CAP * KP:(B + 1) * CAP * KP], non_blocking=True)
            d_desc[:CAP * 32].copy_(d_desc[B * CAP * 32:(B + 1) * CAP * 32], non_blocking=True)
            d_cnt[:1].copy_(d_cnt[B:B + 1], non_blocking=True)

        def timeit(fn):
            for _ in range(5):
                fn()
            torch.cuda.synchronize()
            best = 1e9
            for _ in range(5):
                t0 = time.perf_counter()
                for _ in range(args.steps):
                    fn()
                torch.cuda.synchronize()
                best = min(best, (time.perf_counter() - t0) / args.steps * 1e3)
            return best

        d_kp.zero_(); d_desc.zero_(); d_cnt.zero_()
        step(); step()
        torch.cuda.synchronize()
        sig = (d_nout.cpu().numpy().copy(), d_out.cpu().numpy().copy(), d_cnt.cpu().numpy().copy())
        if ref is None:
            ref = sig
        same = all(np.array_equal(a, b) for a, b in zip(ref, sig))
        eager = timeit(step)
        cap_s = torch.cuda.Stream(device=dev)
        cap_s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(cap_s):
            step()
        torch.cuda.current_stream().wait_stream(cap_s)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=cap_s, capture_error_mode="thread_local"):
            step()
        graph = timeit(g.replay)
        print("ways %d match_joined %d: eager %.4f ms/step, graph %.4f ms/step (%.0f frames/s), same results: %s"
              % (ways, args.match_joined, eager, graph, B / graph * 1e3, same), flush=True)
        del g, exs, ms


if __name__ == "__main__":
    main()

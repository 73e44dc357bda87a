#!/usr/bin/env python3
"""Benchmark of the MI355X ORB frontend hot path.

Primary line (one JSON object on stdout, rank 0): frames/s of ORB extraction +
brute-force Hamming matching against the previous frame at 640x480, 2000
features, 8 levels x 1.2 -- the headline benchmark configuration.  A step is one batch of
frames already resident in HBM: extract the batch, then match frame i against
frame i-1 (the first frame of a batch against the last frame of the batch
before).  With N > 1 every rank processes its own stream of frames (weak
scaling, no data-path collective).

The same line carries:
  roofline      the dominant kernel of the timed region, timed with HIP events
                on the stream it runs on
  cpu_baseline  the CPU oracle (oracle/, 1 thread) on a bounded sample, rank 0, N = 1
  extra         keyframe-pair matching (the multi-GPU benchmark configuration): all pairs of
                --pair-keyframes x 2000 descriptors sharded over the ranks, results
                gathered to rank 0 over RCCL, in Gcmp/s
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E peak
# Integer ceiling of a 256-bit Hamming compare on gfx950, measured with tools/valu_peak.hip
# (8 v_xor + 8 chained v_bcnt, operands in registers, nothing else running): v_bcnt issues at
# half rate, so the ceiling is far below the 4914 Gcmp/s of "16 full-rate ops per compare".
VALU_PEAK_GCMP = 2638.0
VALU_NOMINAL_GCMP = 4914.0     # 256 CU x 128 lanes/clk x 2.4 GHz / 16 ops, for reference only


def pyramid_pixels(ws, hs):
    return [int(w) * int(h) for w, h in zip(ws, hs)]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=64, help="frames per step and GPU")
    ap.add_argument("--width", type=int, default=640)
    ap.add_argument("--height", type=int, default=480)
    ap.add_argument("--nfeatures", type=int, default=2000)
    ap.add_argument("--capacity", type=int, default=4096)
    ap.add_argument("--pair-keyframes", type=int, default=1024)
    ap.add_argument("--pair-desc", type=int, default=2000)
    ap.add_argument("--pair-chunks", type=int, default=16)
    ap.add_argument("--no-pairs", action="store_true")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--cpu-frames", type=int, default=60)
    ap.add_argument("--workload", default="frames", choices=["frames", "stream4k", "projection"],
                    help="frames: 640x480 / 2000 features (default, the headline line); stream4k: 3840x2160 stream; "
                         "projection: 1241x376, 20000 map points in a 50 px window")
    ap.add_argument("--stream-frames", type=int, default=300)
    ap.add_argument("--no-graph", action="store_true", help="time eager launches instead of a HIP graph replay")
    args = ap.parse_args()
    if args.workload == "stream4k":
        return run_stream4k(args)
    if args.workload == "projection":
        return run_projection(args)

    import torch
    import torch.distributed as dist
    from slam_amd import capi, synth

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus %d needs a torch.distributed.run launch with that many ranks" % args.gpus)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=dev)

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    W, H, NF, B, CAP = args.width, args.height, args.nfeatures, args.batch, args.capacity
    params = capi.orb_params(nfeatures=NF)
    ex = capi.Extractor(params, W, H, max_batch=B, max_keypoints=CAP, device=local_rank)
    m = capi.Matcher(device=local_rank)
    m.reserve(B * CAP)
    ws, hs, quotas, scales = ex.levels()

    # ---- inputs: a drifting sequence of one synthetic scene, resident in HBM ----
    n_distinct = min(B, 8)
    seq = synth.frame_sequence(W, H, n_distinct, seed=1 + rank)
    frames = np.stack([seq[i % n_distinct] for i in range(B)])
    d_img = torch.from_numpy(frames).to(dev)
    KP, DM = 28, 16
    d_kp = torch.zeros((B + 1) * CAP * KP, dtype=torch.uint8, device=dev)
    d_desc = torch.zeros((B + 1) * CAP * 32, dtype=torch.uint8, device=dev)
    d_cnt = torch.zeros(B + 1, dtype=torch.int32, device=dev)
    d_out = torch.zeros(B * CAP * DM, dtype=torch.uint8, device=dev)
    d_nout = torch.zeros(B, dtype=torch.int32, device=dev)
    stream = torch.cuda.current_stream().cuda_stream
    kp_p, desc_p, cnt_p = d_kp.data_ptr(), d_desc.data_ptr(), d_cnt.data_ptr()

    def step():
        nonlocal stream
        ex.extract_batch_async(d_img, B, H * W, W, kp_p + CAP * KP, desc_p + CAP * 32, cnt_p + 4, CAP, stream)
        m.frame_match_batch_async(B, desc_p, kp_p, CAP, cnt_p, desc_p + CAP * 32, kp_p + CAP * KP, CAP,
                                  cnt_p + 4, d_out, d_nout, max_dist=300.0, area=-1.0, max_level=-1,
                                  stream=stream)
        # the last frame of this batch is the "previous frame" of the next one
        d_kp[:CAP * KP].copy_(d_kp[B * CAP * KP:(B + 1) * CAP * KP], non_blocking=True)
        d_desc[:CAP * 32].copy_(d_desc[B * CAP * 32:(B + 1) * CAP * 32], non_blocking=True)
        d_cnt[:1].copy_(d_cnt[B:B + 1], non_blocking=True)

    step()  # fills slot 0
    for _ in range(args.warmup):
        step()
    # ---- eager pass with per-stage HIP events (stage times, roofline of the dominant kernel) ----
    capi.prof_enable(args.steps * 12 + 16)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    dt_eager = time.perf_counter() - t0
    stages = capi.prof_collect()
    capi.prof_enable(0)
    # ---- timed region: the same step captured once into a HIP graph and replayed ----
    graph = None
    # With a process group up (N > 1) the launches stay eager: RCCL's watchdog thread polls
    # events, which a stream capture in global mode on another thread does not tolerate.
    if not args.no_graph and world == 1:
        try:
            torch.cuda.synchronize()
            side = torch.cuda.Stream(device=dev)
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                stream = side.cuda_stream
                step()                      # warm-up on the capture stream
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph, stream=side):
                stream = side.cuda_stream
                step()
            stream = torch.cuda.current_stream().cuda_stream
            for _ in range(args.warmup):
                graph.replay()
        except Exception as e:  # capture unsupported: report the eager timing
            sys.stderr.write("graph capture failed (%s); timing eager launches\n" % e)
            graph = None
            stream = torch.cuda.current_stream().cuda_stream
    if graph is not None:
        barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            graph.replay()
        barrier()
        dt = time.perf_counter() - t0
    else:
        # eager launches, timed again without the per-stage events of the pass above
        barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        barrier()
        dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    counts = d_cnt[1:].cpu().numpy()
    nmatch = d_nout.cpu().numpy()
    frames_per_s = world * B * args.steps / dt

    # ---- latency of one frame at a time (batch 1), same context ----
    lat_ms = None
    lat_graph_ms = None
    if rank == 0:
        for _ in range(3):
            ex.extract_batch_async(d_img, 1, H * W, W, kp_p + CAP * KP, desc_p + CAP * 32, cnt_p + 4, CAP, stream)
            m.frame_match_batch_async(1, desc_p, kp_p, CAP, cnt_p, desc_p + CAP * 32, kp_p + CAP * KP, CAP,
                                      cnt_p + 4, d_out, d_nout, 300.0, -1.0, -1, stream)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        reps = 50
        for _ in range(reps):
            ex.extract_batch_async(d_img, 1, H * W, W, kp_p + CAP * KP, desc_p + CAP * 32, cnt_p + 4, CAP, stream)
            m.frame_match_batch_async(1, desc_p, kp_p, CAP, cnt_p, desc_p + CAP * 32, kp_p + CAP * KP, CAP,
                                      cnt_p + 4, d_out, d_nout, 300.0, -1.0, -1, stream)
            torch.cuda.synchronize()
        lat_ms = (time.perf_counter() - t1) / reps * 1e3
        lat_graph_ms = None
        if graph is not None:
            try:
                def one():
                    ex.extract_batch_async(d_img, 1, H * W, W, kp_p + CAP * KP, desc_p + CAP * 32, cnt_p + 4, CAP, stream)
                    m.frame_match_batch_async(1, desc_p, kp_p, CAP, cnt_p, desc_p + CAP * 32, kp_p + CAP * KP, CAP,
                                              cnt_p + 4, d_out, d_nout, 300.0, -1.0, -1, stream)
                side = torch.cuda.Stream(device=dev)
                side.wait_stream(torch.cuda.current_stream())
                g1 = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g1, stream=side):
                    stream = side.cuda_stream
                    one()
                stream = torch.cuda.current_stream().cuda_stream
                for _ in range(3):
                    g1.replay()
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                for _ in range(reps):
                    g1.replay()
                    torch.cuda.synchronize()
                lat_graph_ms = (time.perf_counter() - t1) / reps * 1e3
            except Exception as e:
                sys.stderr.write("batch-1 graph capture failed: %s\n" % e)
                stream = torch.cuda.current_stream().cuda_stream

    # ---- roofline of the dominant kernel (algorithmic bytes per launch / event time) ----
    px = pyramid_pixels(ws, hs)
    P, P0, Ptop = sum(px), px[0], px[-1]
    K = float(counts.mean())
    alg_bytes = {
        "resize_level (all levels)": B * ((P - Ptop) + (P - P0)),
        # the fused kernel covers two stages of the un-fused accounting:
        # FAST (P read) + 7x7 blur (P read + P written)
        "fast_nms": B * 3 * P,
        "describe": B * K * (749 + 512 + 28 + 32),
    }
    roofline = None
    kernel_stages = {k: v for k, v in stages.items() if k != "memset"}
    if kernel_stages:
        dom = max(kernel_stages, key=lambda k: kernel_stages[k][0])
        ms, calls = kernel_stages[dom]
        avg_s = ms / calls * 1e-3
        if dom in alg_bytes:
            ach = alg_bytes[dom] / avg_s / 1e9
            roofline = {"kernel": dom, "bound": "hbm", "achieved": round(ach, 2), "peak": HBM_PEAK_GBS,
                        "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 5),
                        "traffic": pmc_traffic(dom, B, W, H),
                        "avg_launch_us": round(avg_s * 1e6, 2),
                        "algorithmic_bytes_per_launch": int(alg_bytes[dom])}
        else:
            # matching is bound by integer VALU throughput (xor + popcount), not HBM or MFMA
            cmp_per_launch = float((counts.astype(np.float64) * np.roll(counts, 1)).sum())
            ach = cmp_per_launch / avg_s / 1e9
            roofline = {"kernel": dom, "bound": "valu", "achieved": round(ach, 2), "peak": VALU_PEAK_GCMP,
                        "unit": "Gcmp/s", "frac": round(ach / VALU_PEAK_GCMP, 5), "traffic": None,
                        "avg_launch_us": round(avg_s * 1e6, 2)}
    stage_us = {k: round(v[0] / v[1] * 1e3, 2) for k, v in stages.items()}

    # ---- keyframe-pair matching, sharded over ranks, gathered to rank 0 ----
    extra = {"launch_mode": "hip graph replay" if graph is not None else "eager",
             "ms_per_step_eager_with_stage_events": round(dt_eager / args.steps * 1e3, 4),
             "stage_us_per_launch": stage_us, "latency_ms_batch1": None if lat_ms is None else round(lat_ms, 4),
             "latency_ms_batch1_graph": None if lat_graph_ms is None else round(lat_graph_ms, 4),
             "keypoints_per_frame": round(K, 1), "matches_per_frame": round(float(nmatch.mean()), 1)}
    if not args.no_pairs:
        extra["pairs"] = run_pairs(args, torch, dist, capi, synth, m, dev, rank, world, barrier)

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu:
        cpu = cpu_baseline(args, synth, seq)

    if world > 1:
        dist.destroy_process_group()
    if rank == 0:
        line = {
            "metric": "frames/s ORB extract+match @640x480 2k feats",
            "value": round(frames_per_s, 2), "unit": "frames/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u8", "data": "synthetic",
            "config": {"workload": "%dx%d synthetic frame stream, %d ORB features, 8-level 1.2 pyramid, "
                                   "brute-force Hamming vs previous frame (gates off)" % (W, H, NF),
                       "frames_per_step_per_gpu": B, "keypoint_capacity": CAP,
                       "parallelism": "replicas" if world > 1 else "single"},
            "roofline": roofline, "cpu_baseline": cpu, "extra": extra,
        }
        print(json.dumps(line), flush=True)


def pmc_traffic(kernel, batch, width, height):
    """HBM-side bytes per launch of `kernel` from the committed rocprofv3 PMC passes
    (profiles/pmc_traffic.json, written by tools/summarize_profile.py from separate FETCH_SIZE
    and WRITE_SIZE runs of this same command): 2 x FETCH_SIZE (gfx950 counts dword-per-lane
    reads at half their bytes) + WRITE_SIZE.  None when no pass matches this batch and size."""
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "pmc_traffic.json")
    try:
        with open(path) as f:
            rec = json.load(f)
    except (OSError, ValueError):
        return None
    k = rec.get("kernels", {}).get(kernel.split(" ")[0])
    if not k or rec.get("batch") != batch or rec.get("width") != width or rec.get("height") != height:
        return None
    return int(2 * k["fetch_size_kib"] * 1024 + k["write_size_kib"] * 1024)


def run_pairs(args, torch, dist, capi, synth, m, dev, rank, world, barrier):
    """All i<j pairs of n_kf keyframes x n_per descriptors, gates off (the multi-GPU benchmark configuration)."""
    from slam_amd import pairs as pairs_mod
    n_kf, n_per = args.pair_keyframes, args.pair_desc
    desc = synth.random_descriptors(n_kf * n_per, seed=3)
    d_desc = torch.from_numpy(desc).to(dev)
    pairs = pairs_mod.all_pairs(n_kf)
    n_pairs = len(pairs)
    pg = pairs_mod.PairGather(torch, dist, n_pairs, n_per, world, rank, dev, args.pair_chunks)
    d_pairs = torch.from_numpy(np.ascontiguousarray(pairs[pg.lo:max(pg.hi, pg.lo + 1)])).to(dev)
    stream = torch.cuda.current_stream().cuda_stream

    def compute(a, b):
        m.match_pairs_async(d_desc, None, n_kf, n_per, d_pairs.data_ptr() + a * 8, b - a,
                            pg.res.data_ptr() + a * n_per * 4, 300.0, -1.0, -1, stream)

    pg.run(compute, limit=64)  # warm-up on a few pairs per chunk, no gather
    barrier()
    t0 = time.perf_counter()
    pg.run(compute)
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    total_cmp = float(n_pairs) * n_per * n_per
    res = {"workload": "%d keyframes x %d descriptors, all %d pairs i<j, gates off, results gathered to rank 0"
                       % (n_kf, n_per, n_pairs),
           "gcmp_per_s": round(total_cmp / dt / 1e9, 1), "seconds": round(dt, 4), "n_gpus": world,
           "scaling": "strong", "valu_peak_gcmp_per_gpu": VALU_PEAK_GCMP,
           "valu_nominal_gcmp_per_gpu": VALU_NOMINAL_GCMP,
           "valu_frac": round(total_cmp / dt / 1e9 / (VALU_PEAK_GCMP * world), 4)}
    if rank == 0:
        # order-independent checksum of the gathered result, comparable across N
        out = pg.result()
        res["checksum"] = int(out.to(torch.int64).sum().item())
    return res


def run_stream4k(args):
    """4K stream configuration: 3840x2160 stream, 10000 features, extract + match vs previous frame,
    frames arriving in pinned host memory, H2D / compute / D2H overlapped on three streams."""
    import torch
    from slam_amd import capi, synth
    from slam_amd.stream import FramePipeline
    W, H, NF, CAP = 3840, 2160, 10000, 12288
    n = args.stream_frames
    seq = synth.frame_sequence(W, H, 4, seed=1)
    pipe = FramePipeline(torch, W, H, capi.orb_params(nfeatures=NF), CAP, depth=4)
    for i in range(pipe.depth):                 # a producer would write here; pre-filled once
        pipe.h_img[i].numpy()[...] = seq[i % len(seq)]
    for i in range(8):                          # warm-up
        pipe.slot_image(i); pipe.submit(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(8, 8 + n):
        pipe.slot_image(i)                      # waits until the slot's previous results are out
        pipe.submit(i)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    kp, desc, matches = pipe.result(8 + n - 1)
    # the same work with the frames already resident in HBM and no result copies
    pipe2 = FramePipeline(torch, W, H, capi.orb_params(nfeatures=NF), CAP, depth=4, copy_results=False)
    d_frames = [torch.from_numpy(seq[i % len(seq)]).cuda() for i in range(4)]
    st = torch.cuda.current_stream().cuda_stream

    def resident_step(i):
        s_, p_ = i % 4, (i - 1) % 4
        pipe2.ex.extract_batch_async(d_frames[s_], 1, 0, W, pipe2.d_kp[s_], pipe2.d_desc[s_], pipe2.d_cnt[s_], CAP, st)
        pipe2.m.frame_match_async(pipe2.d_desc[p_], pipe2.d_kp[p_], CAP, pipe2.d_cnt[p_], pipe2.d_desc[s_],
                                  pipe2.d_kp[s_], CAP, pipe2.d_cnt[s_], pipe2.d_match[s_], pipe2.d_nmatch[s_],
                                  max_dist=300.0, area=-1.0, max_level=-1, stream=st)
    for i in range(8):
        resident_step(i)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    for i in range(8, 8 + n):
        resident_step(i)
    torch.cuda.synchronize()
    dt2 = time.perf_counter() - t1
    pipe.close(); pipe2.close()
    print(json.dumps({
        "metric": "frames/s ORB extract+match @3840x2160 10k feats, pipelined stream", "unit": "frames/s",
        "value": round(n / dt2, 2), "value_including_h2d_and_d2h": round(n / dt, 2),
        "n_gpus": 1, "steps": n, "ms_per_frame_resident": round(dt2 / n * 1e3, 4),
        "ms_per_frame_pipelined": round(dt / n * 1e3, 4),
        "h2d_bytes_per_frame": W * H, "d2h_bytes_per_frame": CAP * (28 + 32 + 16) + 8,
        "keypoints_last_frame": len(kp), "matches_last_frame": len(matches),
        "config": {"workload": "3840x2160 synthetic stream, 10000 ORB features, match vs previous frame, "
                               "frames in pinned host memory, 3 streams, 4 slots"},
        "data": "synthetic", "dtype": "u8"}), flush=True)


def run_projection(args):
    """KITTI-shaped configuration: 1241x376 frames, 4000 features; 20000 map-point descriptors with
    projected positions matched against the frame's keypoints, window 50 px, maxLevel 4."""
    import torch
    from slam_amd import capi, synth
    W, H, NF, CAP, NQ = 1241, 376, 4000, 8192, 20000
    dev = torch.device("cuda", 0)
    ex = capi.Extractor(capi.orb_params(nfeatures=NF), W, H, max_batch=1, max_keypoints=CAP)
    m = capi.Matcher()
    m.reserve(NQ)
    seq = synth.frame_sequence(W, H, 4, seed=2)
    d_frames = [torch.from_numpy(f).cuda() for f in seq]
    d_kp = torch.zeros(CAP * 28, dtype=torch.uint8, device=dev)
    d_desc = torch.zeros(CAP * 32, dtype=torch.uint8, device=dev)
    d_cnt = torch.zeros(1, dtype=torch.int32, device=dev)
    # local map: 20000 points whose descriptors are noisy copies of frame descriptors
    kp0, desc0 = ex.extract(seq[0])
    rng = np.random.default_rng(0)
    src = rng.integers(0, len(kp0), NQ)
    q_desc = torch.from_numpy(synth.noisy_copy(desc0[src], 20, 1)).cuda()
    q_kp_np = kp0[src].copy()
    q_kp = torch.from_numpy(q_kp_np.view(np.uint8).copy()).cuda()
    q_xy = torch.from_numpy(np.stack([q_kp_np["x"] + rng.normal(0, 8, NQ), q_kp_np["y"] + rng.normal(0, 8, NQ)],
                                     1).astype(np.float32)).cuda()
    ids = torch.arange(NQ, dtype=torch.int32, device=dev)
    d_out = torch.zeros(NQ * 16, dtype=torch.uint8, device=dev)
    d_n = torch.zeros(1, dtype=torch.int32, device=dev)
    st = torch.cuda.current_stream().cuda_stream

    def step(i):
        ex.extract_batch_async(d_frames[i % 4], 1, 0, W, d_kp, d_desc, d_cnt, CAP, st)
        m.frame_match_async(q_desc, q_kp, NQ, None, d_desc, d_kp, CAP, d_cnt, d_out, d_n, d_ids=ids, n_ids=NQ,
                            d_q_xy=q_xy, max_dist=300.0, area=50.0, max_level=4, stream=st)
    for i in range(args.warmup):
        step(i)
    capi.prof_enable(args.steps * 12 + 16)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(i)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    stages = capi.prof_collect()
    n_t = int(d_cnt.item())
    print(json.dumps({
        "metric": "frames/s ORB extract + projection-windowed match @1241x376 4k feats vs 20k map points",
        "unit": "frames/s", "value": round(args.steps / dt, 2), "n_gpus": 1, "steps": args.steps,
        "ms_per_step": round(dt / args.steps * 1e3, 4), "keypoints": n_t, "matches": int(d_n.item()),
        "match_stage_us": round(sum(stages[k][0] / stages[k][1] for k in ("match", "spatial_sort", "match_window")
                                    if k in stages) * 1e3, 2),
        "equivalent_full_sweep_gcmp_per_s": round(NQ * n_t / sum(
            stages[k][0] / stages[k][1] * 1e-3 for k in ("match", "spatial_sort", "match_window") if k in stages) / 1e9, 1),
        "stage_us_per_launch": {k: round(v[0] / v[1] * 1e3, 2) for k, v in stages.items()},
        "data": "synthetic", "dtype": "u8"}), flush=True)
    ex.close(); m.close()


def cpu_baseline(args, synth, seq):
    """The CPU oracle (1 thread) on a bounded sample of the same workload."""
    from oracle import orb_oracle as oracle
    po = oracle.params(nfeatures=args.nfeatures)
    n = args.cpu_frames
    prev = oracle.detect_and_compute(po, seq[0])
    t0 = time.perf_counter()
    for i in range(n):
        cur = oracle.detect_and_compute(po, seq[(i + 1) % len(seq)])
        oracle.match_gated(prev[1], prev[0], cur[1], cur[0], 300.0, -1.0, -1)
        prev = cur
    dt = time.perf_counter() - t0
    out = {"value": round(n / dt, 3), "unit": "frames/s", "cores": 1, "kind": "port",
           "sample": "%d frames of the same %dx%d / %d-feature stream: extract + brute-force match vs previous "
                     "frame, CPU oracle (oracle/orb_oracle.c, gcc -O2, 1 thread as the reference runs)" %
                     (n, args.width, args.height, args.nfeatures)}
    # the same oracle with OpenMP over frames (extraction) and queries (matching)
    threads = oracle.max_threads()
    if threads > 1:
        m = 4 * threads
        frames = np.stack([seq[i % len(seq)] for i in range(m + 1)])
        t0 = time.perf_counter()
        res = oracle.detect_and_compute_batch(po, frames)
        for i in range(m):
            oracle.match_gated_mt(res[i][1], res[i][0], res[i + 1][1], res[i + 1][0], 300.0, -1.0, -1)
        dt = time.perf_counter() - t0
        out["all_cores"] = {"value": round(m / dt, 3), "unit": "frames/s", "cores": threads,
                            "sample": "%d frames, OpenMP over frames and queries" % m}
    return out


if __name__ == "__main__":
    try:
        main()
    except BaseException as e:
        if isinstance(e, SystemExit) and not e.code:
            raise
        # Leave without unwinding GPU objects: destructors that run after a failed HIP call
        # can throw again and turn an ordinary error into an abort with work still queued.
        import traceback
        traceback.print_exc()
        sys.stderr.flush()
        sys.stdout.flush()
        os._exit(1)

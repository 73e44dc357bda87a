#!/usr/bin/env python3
"""Benchmark of the MI355X ORB frontend hot path.

Primary line (one JSON object on stdout, rank 0): frames/s of ORB extraction +
brute-force Hamming matching against the previous frame at 640x480, 2000
features, 8 levels x 1.2 -- the headline benchmark configuration.  A step is one batch of
frames already resident in HBM: extract the batch, then match frame i against
frame i-1 (the first frame of a batch against the last frame of the batch
before).  With N > 1 every rank processes its own stream of frames (weak
scaling, no data-path collective).

`python bench.py --gpus N` starts its own N ranks (one process per GPU) when it was not
started by torch.distributed.run; under torch.distributed.run it is one of the ranks.

The same line carries:
  roofline      the dominant kernel of the timed region, timed with HIP events
                on the stream it runs on
  cpu_baseline  the CPU oracle (oracle/, 1 thread) on a bounded sample, rank 0, N = 1
  extra         the same step with the ratio test on; the same step with frames arriving in
                pinned host memory (H2D and D2H overlapped with compute); keyframe-pair
                matching (the multi-GPU benchmark configuration): all pairs of
                --pair-keyframes x 2000 descriptors sharded over the ranks, results
                gathered to rank 0 over RCCL, in Gcmp/s
"""
import argparse
import json
import os
import socket
import statistics
import subprocess
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E peak
CLOCK_GHZ = 2.4
N_SIMD = 1024                  # 256 CUs x 4 SIMD-32; a wave64 VALU instruction holds one for 2 cycles
VALU_NOMINAL_GCMP = 4914.0     # 256 CU x 128 lanes/clk x 2.4 GHz / 16 ops, for reference only
# matrix-core ceiling of the matcher: a 32 x 32 block of (train, query) pairs costs 4
# v_mfma_scale_f32_32x32x64_f8f6f4 on 4-bit operands, 32 cycles each on one of the 1024 SIMDs
MFMA_F4_PEAK_GCMP = N_SIMD * 2.4e9 * 1024 / (4 * 32) / 1e9


def valu_peak_gcmp():
    """Integer ceiling of a 256-bit Hamming compare on gfx950 in Gcmp/s, read from the committed
    output of tools/valu_peak.hip (8 v_xor + 8 chained v_bcnt per compare, operands in registers,
    nothing else running): v_bcnt issues at half rate, so the ceiling is far below the nominal
    "16 full-rate ops per compare" figure."""
    path = os.path.join(ROOT, "profiles", "valu_peak.json")
    with open(path) as f:
        return float(json.load(f)["hamming256_gcmp_per_s"])


def pyramid_pixels(ws, hs):
    return [int(w) * int(h) for w, h in zip(ws, hs)]


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--repeats", type=int, default=5,
                    help="the timed region (exactly --steps steps) is run this many times; the line reports the median")
    ap.add_argument("--batch", type=int, default=128,
                    help="frames per step and GPU (128: the pyramid's workgroups fill the chip three times exactly and the blurred "
                         "levels of a batch still fit the Infinity Cache; 64 is 8 % slower per frame, 256 no faster)")
    ap.add_argument("--width", type=int, default=640)
    ap.add_argument("--height", type=int, default=480)
    ap.add_argument("--nfeatures", type=int, default=2000)
    ap.add_argument("--capacity", type=int, default=4096)
    ap.add_argument("--pair-keyframes", type=int, default=1024)
    ap.add_argument("--pair-desc", type=int, default=2000)
    ap.add_argument("--pair-chunks", type=int, default=16)
    ap.add_argument("--pair-deadline", type=float, default=240.0,
                    help="seconds after which the line is printed without the keyframe-pair leg")
    ap.add_argument("--no-pairs", action="store_true")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the ratio-test and host-pipelined variants")
    ap.add_argument("--cpu-frames", type=int, default=60)
    ap.add_argument("--workload", default="frames", choices=["frames", "stream4k", "projection"],
                    help="frames: 640x480 / 2000 features (default, the headline line); stream4k: 3840x2160 stream; "
                         "projection: 1241x376, 20000 map points in a 50 px window")
    ap.add_argument("--stream-frames", type=int, default=1000)
    ap.add_argument("--stream-batches", default="1,2,4",
                    help="stream4k: frames queued per launch sequence, one timed run each; `value` is the last one's")
    ap.add_argument("--no-graph", action="store_true", help="time eager launches instead of a HIP graph replay")
    ap.add_argument("--graph-order", default="late", choices=["early", "late"],
                    help="early: both graphs (batch, single frame) are captured and instantiated before either is "
                         "launched; late: each graph is captured right before its first replay")
    ap.add_argument("--rehearse-cpu", action="store_true",
                    help="launcher / rendezvous / gather rehearsal on the CPU over gloo: no GPU work, synthetic result rows")
    return ap.parse_args(argv)


# ------------------------------------------------------------------------------------------------
# Launcher: one process per GPU, started before anything in this process touches the GPU
# ------------------------------------------------------------------------------------------------
def launch_ranks(args, argv):
    n = args.gpus
    if not args.rehearse_cpu:
        import torch  # device_count() does not initialise the GPU
        have = torch.cuda.device_count()
        if have < n:
            sys.stderr.write("bench.py: --gpus %d needs %d GPUs, %d visible\n" % (n, n, have))
            return 2
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(n):
        env = dict(os.environ)
        env.update({"RANK": str(r), "LOCAL_RANK": str(r), "WORLD_SIZE": str(n), "LOCAL_WORLD_SIZE": str(n),
                    "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port)})
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        out = None if r == 0 else subprocess.DEVNULL      # rank 0 prints the line on our stdout
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env, stdout=out))
    rc = 0
    alive = set(range(n))
    while alive:
        for r in list(alive):
            code = procs[r].poll()
            if code is None:
                continue
            alive.discard(r)
            if code != 0 and rc == 0:
                rc = code if code > 0 else 1
                sys.stderr.write("bench.py: rank %d exited with %d; stopping the other ranks\n" % (r, code))
                for o in alive:
                    procs[o].terminate()
        time.sleep(0.05)
    return rc


class Ranks:
    """Rank bookkeeping and the control-plane group (gloo: barriers, timing reductions, handing the
    RCCL unique id of the pair shard to the other ranks)."""

    def __init__(self, torch, dist, rehearse):
        self.torch, self.dist = torch, dist
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.rank = int(os.environ.get("RANK", "0"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        self.rehearse = rehearse
        if self.world > 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            # Control plane on gloo: the frame workload has no data-path collective, and a process
            # group without a GPU watchdog thread leaves HIP graph capture undisturbed.
            # (gloo announces its connections on stdout: keep stdout for the one JSON line)
            sys.stdout.flush()
            saved = os.dup(1)
            os.dup2(2, 1)
            try:
                dist.init_process_group("gloo", rank=self.rank, world_size=self.world)
                dist.barrier()
            finally:
                sys.stdout.flush()
                os.dup2(saved, 1)
                os.close(saved)

    def barrier(self):
        if not self.rehearse:
            self.torch.cuda.synchronize()
        if self.world > 1:
            self.dist.barrier()
        if not self.rehearse:
            self.torch.cuda.synchronize()

    def max_over_ranks(self, values):
        if self.world == 1:
            return list(values)
        t = self.torch.tensor(list(values), dtype=self.torch.float64)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return [float(v) for v in t]

    def data_plane(self, dev):
        """CPU rehearsal only: the group that carries the result rows there (gloo).  On the GPUs the
        C++ pair shard owns its RCCL communicator; torch creates none, so no RCCL watchdog thread ever
        runs beside a graph capture."""
        return self.dist.group.WORLD if self.world > 1 else None

    def close(self):
        if self.world > 1:
            self.dist.barrier()
            self.dist.destroy_process_group()


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    args = parse_args(argv)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return launch_ranks(args, argv)
    if args.workload == "stream4k":
        return run_stream4k(args)
    if args.workload == "projection":
        return run_projection(args)
    if args.rehearse_cpu:
        return run_rehearsal(args)
    return run_frames(args)


def timed_repeats(ranks, fn, steps, repeats):
    """`repeats` timed regions of exactly `steps` calls of fn, each between two barriers; every
    region's time is the maximum over the ranks.  Returns the list of region times (seconds)."""
    times = []
    for _ in range(max(1, repeats)):
        ranks.barrier()
        t0 = time.perf_counter()
        for _ in range(steps):
            fn()
        ranks.barrier()
        times.append(time.perf_counter() - t0)
    return ranks.max_over_ranks(times)


def run_frames(args):
    import torch
    import torch.distributed as dist
    from slam_amd import capi, synth

    ranks = Ranks(torch, dist, False)
    world, rank, local_rank = ranks.world, ranks.rank, ranks.local_rank
    if args.gpus != world:
        raise SystemExit("--gpus %d does not match WORLD_SIZE %d" % (args.gpus, world))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    W, H, NF, B, CAP = args.width, args.height, args.nfeatures, args.batch, args.capacity
    params = capi.orb_params(nfeatures=NF)
    ex = capi.Extractor(params, W, H, max_batch=B, max_keypoints=CAP, device=local_rank)
    m = capi.Matcher(device=local_rank)
    m.reserve(B, CAP, CAP)                      # before any capture: captured calls never allocate
    m_ratio = capi.Matcher(device=local_rank)   # the same step with the ratio test on (extra)
    m_ratio.set_options(0.8, 0)
    m_ratio.reserve(B, CAP, CAP)
    ws, hs, quotas, scales = ex.levels()

    # ---- inputs: a drifting sequence of one synthetic scene, resident in HBM ----
    seq = synth.frame_sequence(W, H, B, seed=1 + rank)     # B distinct frames
    frames = np.stack(seq)
    d_img = torch.from_numpy(frames).to(dev)
    KP, DM = 28, 16
    # Two sets of result arrays used in turn: a step extracts its batch into one set and matches frame b
    # with frame b - 1 of the same set, frame 0 with the last frame of the step before, which lies in
    # the other set (slam_amd_frame_match_chain_async) -- no record is copied between steps.
    sets = [{"kp": torch.zeros(B * CAP * KP, dtype=torch.uint8, device=dev),
             "desc": torch.zeros(B * CAP * 32, dtype=torch.uint8, device=dev),
             "cnt": torch.zeros(B, dtype=torch.int32, device=dev)} for _ in range(2)]
    d_out = torch.zeros(B * CAP * DM, dtype=torch.uint8, device=dev)
    d_nout = torch.zeros(B, dtype=torch.int32, device=dev)
    cur = {"stream": torch.cuda.current_stream().cuda_stream, "parity": 0}

    class Step:
        """One step; successive calls alternate between the two sets, eager or as a pair of HIP graphs."""

        def __init__(self, matcher, batch, area=-1.0, max_level=-1):
            self.matcher, self.batch, self.graphs = matcher, batch, None
            self.area, self.max_level = area, max_level

        def launch(self, parity):
            st = cur["stream"]
            c, p = sets[parity], sets[1 - parity]
            last = self.batch - 1       # where a step of this batch size leaves its last frame
            ex.extract_batch_async(d_img, self.batch, H * W, W, c["kp"].data_ptr(), c["desc"].data_ptr(),
                                   c["cnt"].data_ptr(), CAP, st)
            self.matcher.frame_match_chain_async(self.batch, p["desc"].data_ptr() + last * CAP * 32,
                                                 p["kp"].data_ptr() + last * CAP * KP, p["cnt"].data_ptr() + 4 * last,
                                                 c["desc"].data_ptr(), c["kp"].data_ptr(), CAP, c["cnt"].data_ptr(),
                                                 d_out, d_nout, max_dist=300.0, area=self.area, max_level=self.max_level, stream=st)

        def __call__(self):
            self.launch(cur["parity"])
            cur["parity"] ^= 1

        def replay(self):
            self.graphs[cur["parity"]].replay()
            cur["parity"] ^= 1

    step = Step(m, B)
    one_frame = Step(m, 1)
    step_ratio = Step(m_ratio, B)
    step_octave = Step(m, B, -1.0, 4)          # frameMatch(..., 300, -1, 4) as Mapper::initialize calls it
    step_window = Step(m, B, 50.0, 4)          # ... and with the 50-pixel window of the tracker's calls

    def capture(fn):
        """fn's launches as HIP graphs, one per set (None if capture is not possible)."""
        try:
            graphs = []
            for parity in (0, 1):
                torch.cuda.synchronize()
                side = torch.cuda.Stream(device=dev)
                side.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(side):
                    cur["stream"] = side.cuda_stream
                    fn.launch(parity)           # warm-up on the capture stream
                torch.cuda.current_stream().wait_stream(side)
                torch.cuda.synchronize()
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, stream=side, capture_error_mode="thread_local"):
                    cur["stream"] = side.cuda_stream
                    fn.launch(parity)
                graphs.append(g)
            fn.graphs = graphs
            return fn
        except Exception as e:  # capture unsupported: the eager timing is reported
            sys.stderr.write("graph capture failed (%s); timing eager launches\n" % e)
            return None
        finally:
            cur["stream"] = torch.cuda.current_stream().cuda_stream

    step()  # the other set holds a frame before the first timed step
    for _ in range(args.warmup):
        step()
    for _ in range(3):
        one_frame()
    want_graph = not args.no_graph
    graph = graph1 = None
    if want_graph and args.graph_order == "early":
        graph = capture(step)
        graph1 = capture(one_frame) if rank == 0 else None
    # ---- eager pass with per-stage HIP events (stage times, roofline of the dominant kernel) ----
    capi.prof_enable(args.steps * 12 + 16)
    ranks.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    ranks.barrier()
    dt_eager = time.perf_counter() - t0
    stages = capi.prof_collect()
    capi.prof_enable(0)
    # ---- timed region: the same step replayed from its HIP graph, `repeats` times `steps` steps ----
    if want_graph and graph is None:
        graph = capture(step)
    timed = graph.replay if graph is not None else step
    for _ in range(args.warmup):
        timed()
    times = timed_repeats(ranks, timed, args.steps, args.repeats)
    dt = statistics.median(times)
    counts = sets[cur["parity"] ^ 1]["cnt"].cpu().numpy()
    nmatch = d_nout.cpu().numpy()
    frames_per_s = world * B * args.steps / dt

    # ---- latency of one frame at a time (batch 1), same context ----
    lat_ms = lat_graph_ms = None
    if rank == 0:
        reps = 50
        for _ in range(3):
            one_frame()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(reps):
            one_frame()
            torch.cuda.synchronize()
        lat_ms = (time.perf_counter() - t1) / reps * 1e3
        if want_graph and graph1 is None:
            graph1 = capture(one_frame)
        if graph1 is not None:
            for _ in range(3):
                graph1.replay()
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(reps):
                graph1.replay()
                torch.cuda.synchronize()
            lat_graph_ms = (time.perf_counter() - t1) / reps * 1e3

    # ---- roofline of the dominant kernel (algorithmic bytes per launch / event time) ----
    px = pyramid_pixels(ws, hs)
    P, P0, Ptop = sum(px), px[0], px[-1]
    K = float(counts.mean())
    valu_peak = valu_peak_gcmp()
    alg_bytes = {
        "pyramid": B * ((P - Ptop) + (P - P0)),
        "resize_level": B * ((P - Ptop) + (P - P0)),
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
        key = dom.split(" ")[0]
        if key in alg_bytes:
            ach = alg_bytes[key] / avg_s / 1e9
            pmc = pmc_counters(key, B, W, H)
            roofline = {"kernel": dom, "bound": "hbm", "achieved": round(ach, 2), "peak": HBM_PEAK_GBS,
                        "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 5),
                        "traffic": pmc.get("traffic"), "traffic_source": pmc.get("source"),
                        "avg_launch_us": round(avg_s * 1e6, 2),
                        "algorithmic_bytes_per_launch": int(alg_bytes[key])}
            if pmc.get("valu_wave_insts"):
                # the kernel's other ceiling: VALU issue.  A wave64 VALU instruction holds one of the
                # 1024 SIMDs for 2 cycles if it is one of the few full-rate opcodes and for 4 otherwise
                # (measured: profiles/*valu_ops_stdout.txt).  nominal = every instruction at 2 cycles;
                # priced = the measured instruction count at the kernel's static mix of the two classes
                # (profiles/isa_mix.json, tools/isa_price.py).
                nominal_s = pmc["valu_wave_insts"] * 2.0 / N_SIMD / (CLOCK_GHZ * 1e9)
                roofline["valu_issue"] = {"wave_insts_per_launch": int(pmc["valu_wave_insts"]),
                                          "floor_us": round(nominal_s * 1e6, 2), "frac": round(nominal_s / avg_s, 4)}
                cpi = isa_cycles_per_inst(key)
                if cpi:
                    priced_s = pmc["valu_wave_insts"] * cpi / N_SIMD / (CLOCK_GHZ * 1e9)
                    roofline["valu_issue"].update({"cycles_per_inst_priced": cpi, "priced_floor_us": round(priced_s * 1e6, 2),
                                                   "priced_frac": round(priced_s / avg_s, 4)})
        else:
            # matching is bound by integer VALU throughput (xor + popcount), not HBM or MFMA
            cmp_per_launch = float((counts.astype(np.float64) * np.roll(counts, 1)).sum())
            ach = cmp_per_launch / avg_s / 1e9
            roofline = {"kernel": dom, "bound": "valu", "achieved": round(ach, 2), "peak": valu_peak,
                        "unit": "Gcmp/s", "frac": round(ach / valu_peak, 5), "traffic": None,
                        "avg_launch_us": round(avg_s * 1e6, 2)}
    stage_us = {k: round(v[0] / v[1] * 1e3, 2) for k, v in stages.items()}

    extra = {"launch_mode": "hip graph replay" if graph is not None else "eager",
             "graph_order": args.graph_order if graph is not None else None,
             "timed_regions_ms_per_step": {"median": round(dt / args.steps * 1e3, 4),
                                           "min": round(min(times) / args.steps * 1e3, 4),
                                           "max": round(max(times) / args.steps * 1e3, 4), "repeats": len(times)},
             "ms_per_step_eager_with_stage_events": round(dt_eager / args.steps * 1e3, 4),
             "stage_us_per_launch": stage_us, "latency_ms_batch1": None if lat_ms is None else round(lat_ms, 4),
             "latency_ms_batch1_graph": None if lat_graph_ms is None else round(lat_graph_ms, 4),
             "keypoints_per_frame": round(K, 1), "matches_per_frame": round(float(nmatch.mean()), 1)}

    if not args.no_extras:
        # ---- the same step with the ratio test on (best < 0.8 x second best) ----
        g_ratio = capture(step_ratio) if want_graph else None
        timed_r = g_ratio.replay if g_ratio is not None else step_ratio
        for _ in range(args.warmup):
            timed_r()
        tr = statistics.median(timed_repeats(ranks, timed_r, args.steps, min(3, args.repeats)))
        extra["ratio_test_0.8"] = {"frames_per_s": round(world * B * args.steps / tr, 2),
                                   "ms_per_step": round(tr / args.steps * 1e3, 4),
                                   "matches_per_frame": round(float(d_nout.cpu().numpy().mean()), 1)}
        # ---- the same step with the reference's gates: octave gate (mapper.cpp:34), window + octave (tracker.cpp:74) ----
        for key, st_g in (("octave_gate_4", step_octave), ("window_50_octave_4", step_window)):
            g_g = capture(st_g) if want_graph else None
            timed_g = g_g.replay if g_g is not None else st_g
            for _ in range(args.warmup):
                timed_g()
            tg = statistics.median(timed_repeats(ranks, timed_g, args.steps, min(3, args.repeats)))
            extra[key] = {"frames_per_s": round(world * B * args.steps / tg, 2),
                          "ms_per_step": round(tg / args.steps * 1e3, 4),
                          "vs_gates_off": round(dt / tg, 4),
                          "matches_per_frame": round(float(d_nout.cpu().numpy().mean()), 1)}
        # ---- frames arriving in pinned host memory: H2D, compute and D2H overlapped ----
        extra["host_pipelined"] = run_host_pipelined(args, torch, ranks, frames, world)

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
        "roofline": roofline, "cpu_baseline": None, "extra": extra,
    }
    emitted = threading.Event()

    def emit():
        if rank == 0 and not emitted.is_set():
            emitted.set()
            print(json.dumps(line), flush=True)

    if rank == 0 and world == 1 and not args.no_cpu:
        line["cpu_baseline"] = cpu_baseline(args, synth, seq)

    # ---- keyframe-pair matching, sharded over ranks, gathered to rank 0 ----
    if not args.no_pairs:
        # a rank that hangs in the exchange must not cost the whole line
        def give_up():
            extra["pairs"] = {"error": "not finished within %.0f s" % args.pair_deadline}
            emit()
            sys.stdout.flush()
            os._exit(0)
        guard = threading.Timer(args.pair_deadline, give_up)
        guard.daemon = True
        guard.start()
        try:
            extra["pairs"] = run_pairs(args, torch, dist, capi, synth, m, dev, ranks, valu_peak)
        except Exception as e:      # the frame line stands on its own
            import traceback
            traceback.print_exc()
            extra["pairs"] = {"error": "%s: %s" % (type(e).__name__, e)}
        guard.cancel()
    ex.close(); m.close(); m_ratio.close()
    ranks.close()
    emit()
    return 0


def run_host_pipelined(args, torch, ranks, frames, world):
    """The frame step with inputs in pinned host memory and results copied back, through the library's C++
    pipeline (slam_amd_pipeline_*): three slots of one batch each, three streams (upload of batch i + 1,
    compute of batch i, download of batch i - 1 overlap).  The slots' input buffers are filled once, as a
    camera writing by DMA would leave them; a step = `batch` frames handed in and submitted."""
    from slam_amd import capi
    W, H, B, CAP = args.width, args.height, args.batch, args.capacity
    pipe = capi.Pipeline(capi.orb_params(nfeatures=args.nfeatures), W, H, CAP, depth=3, batch=B, device=ranks.local_rank)
    n = {"i": 0}

    def submit(fill=False):
        for k in range(B):
            i = n["i"]; n["i"] += 1
            buf = pipe.input(i)
            if fill:
                buf[...] = frames[k]
            pipe.submit(i)

    for _ in range(3):
        submit(fill=True)
    submit()
    t = statistics.median(timed_repeats(ranks, submit, args.steps, min(3, args.repeats)))
    kp, desc, mt = pipe.result(n["i"] - 1)
    pipe.close()
    return {"frames_per_s": round(world * B * args.steps / t, 2), "ms_per_step": round(t / args.steps * 1e3, 4),
            "h2d_bytes_per_step": B * W * H, "d2h_bytes_per_step": B * CAP * (28 + 32 + 16) + 8 * B,
            "keypoints_last_frame": int(len(kp)), "matches_last_frame": int(len(mt)),
            "launch_mode": "slam_amd_pipeline_* (C++ host): eager, 3 streams, 3 slots of one batch, pinned host buffers"}


def pmc_counters(kernel, batch, width, height):
    """Per-launch hardware counters of `kernel` from the committed rocprofv3 PMC passes
    (profiles/pmc_counters.json, written by tools/summarize_profile.py from separate FETCH_SIZE,
    WRITE_SIZE and SQ runs of this same command): traffic = 2 x FETCH_SIZE (gfx950 counts
    dword-per-lane reads at half their bytes) + WRITE_SIZE; valu_wave_insts = SQ_INSTS_VALU.
    Empty when no pass matches this batch and image size."""
    path = os.path.join(ROOT, "profiles", "pmc_counters.json")
    try:
        with open(path) as f:
            rec = json.load(f)
    except (OSError, ValueError):
        return {}
    k = rec.get("kernels", {}).get(kernel)
    if not k or rec.get("batch") != batch or rec.get("width") != width or rec.get("height") != height:
        return {}
    out = {"source": "profiles/pmc_counters.json (rocprofv3 --pmc passes of %s)" % rec.get("source_commit", rec.get("tag", "an earlier build"))}
    if "fetch_size_kib" in k and "write_size_kib" in k:
        out["traffic"] = int(2 * k["fetch_size_kib"] * 1024 + k["write_size_kib"] * 1024)
    if "sq_insts_valu" in k:
        out["valu_wave_insts"] = float(k["sq_insts_valu"])
    return out


def isa_cycles_per_inst(kernel):
    """Average issue cycles per VALU instruction of `kernel` at its static mix of full-rate (2-cycle)
    and other (4-cycle) opcodes, from profiles/isa_mix.json; None if the file has no such kernel."""
    try:
        with open(os.path.join(ROOT, "profiles", "isa_mix.json")) as f:
            return json.load(f)["kernels"][kernel]["cycles_per_inst"]
    except (OSError, ValueError, KeyError):
        return None


def run_pairs(args, torch, dist, capi, synth, m, dev, ranks, valu_peak):
    """All i<j pairs of n_kf keyframes x n_per descriptors, gates off (the multi-GPU benchmark configuration).
    Sharding, chunking and the gather to rank 0 are the C++ pair shard's (slam_amd_pair_shard_*, RCCL
    send / recv on its own stream); torch only carries the RCCL unique id to the other ranks."""
    from slam_amd import pairs as pairs_mod
    world, rank = ranks.world, ranks.rank
    n_kf, n_per = args.pair_keyframes, args.pair_desc
    desc = synth.random_descriptors(n_kf * n_per, seed=3)
    d_desc = torch.from_numpy(desc).to(dev)
    pairs = pairs_mod.all_pairs(n_kf)
    n_pairs = len(pairs)
    uid = None
    if world > 1:
        box = [capi.pair_shard_unique_id() if rank == 0 else None]
        dist.broadcast_object_list(box, src=0)
        uid = box[0]
    shard = capi.PairShard(m, ranks.local_rank, uid, world, rank, n_pairs, n_per, args.pair_chunks)
    d_pairs = torch.from_numpy(np.ascontiguousarray(pairs[shard.lo:max(shard.hi, shard.lo + 1)])).to(dev)
    stream = torch.cuda.current_stream().cuda_stream
    shard.run(d_desc, None, n_kf, d_pairs, 300.0, -1.0, -1, stream, limit_rows=64)   # warm-up: a few pairs per chunk, no exchange
    shard.wait(stream)
    ranks.barrier()
    t0 = time.perf_counter()
    shard.run(d_desc, None, n_kf, d_pairs, 300.0, -1.0, -1, stream)
    shard.wait(stream)
    ranks.barrier()
    dt = ranks.max_over_ranks([time.perf_counter() - t0])[0]
    total_cmp = float(n_pairs) * n_per * n_per
    res = {"workload": "%d keyframes x %d descriptors, all %d pairs i<j, gates off, results gathered to rank 0"
                       % (n_kf, n_per, n_pairs),
           "gcmp_per_s": round(total_cmp / dt / 1e9, 1), "seconds": round(dt, 4), "n_gpus": world,
           "scaling": "strong",
           # distances come from v_mfma_scale_f32_32x32x64_f8f6f4 on 4-bit operands: four products of 32 cycles
           # per 32 x 32 block of pairs on each of the 1024 SIMDs (the ceiling is priced at 2.4 GHz; the chip
           # holds about 1.9 GHz under this load)
           "bound": "mfma", "mfma_peak_gcmp_per_gpu": MFMA_F4_PEAK_GCMP,
           "mfma_frac": round(total_cmp / dt / 1e9 / (MFMA_F4_PEAK_GCMP * world), 4),
           "mfma_f4_tmac_per_s": round(total_cmp / dt * 256 / 1e12, 1),
           # the popcount kernel's measured ceiling (used with a gate on), for comparison
           "valu_peak_gcmp_per_gpu": valu_peak,
           "host": "C++ pair shard (slam_amd_pair_shard_*), RCCL send/recv in %d chunks" % args.pair_chunks}
    if rank == 0:
        # order-independent checksum of the gathered result, comparable across N
        class _Table:   # the shard's device table as a torch tensor, without a copy
            __cuda_array_interface__ = {"shape": (n_pairs, n_per), "typestr": "<i4", "data": (shard.result_ptr(), False),
                                        "version": 2, "strides": None}
        d_out = torch.as_tensor(_Table(), device=dev)
        res["checksum"] = int(d_out.to(torch.int64).sum().item())
        # a dozen pairs recomputed with numpy: the rate above belongs to a checked result
        rng = np.random.default_rng(0)
        bad = 0
        sample = rng.integers(0, n_pairs, 12)
        for r in sample:
            i, j = pairs[r]
            bi = np.unpackbits(desc[i * n_per:(i + 1) * n_per], axis=1).astype(np.float32)
            bj = np.unpackbits(desc[j * n_per:(j + 1) * n_per], axis=1).astype(np.float32)
            full = (bi.sum(1)[:, None] + bj.sum(1)[None, :] - 2 * (bi @ bj.T)).astype(np.int32)     # exact: small integers
            want = (full.min(1).astype(np.uint32) << 16) | full.argmin(1).astype(np.uint32)   # argmin = lowest index
            got = d_out[int(r)].cpu().numpy().view(np.uint32)
            bad += int(not np.array_equal(got, want))
        res["checked_pairs"] = {"sampled": len(sample), "differ": bad}
        if bad:
            raise RuntimeError("%d of %d sampled keyframe pairs differ from the numpy recomputation" % (bad, len(sample)))
    shard.close()
    return res


def run_rehearsal(args):
    """CPU rehearsal of everything around the kernels for N ranks: rank start-up, rendezvous,
    barriers, the max-over-ranks reduction and the chunked gather of result rows, over gloo.
    The rows are synthetic (a function of the global pair index); rank 0 checks the gathered
    table and prints one JSON line marked as a rehearsal."""
    import torch
    import torch.distributed as dist
    from slam_amd import pairs as pairs_mod
    ranks = Ranks(torch, dist, True)
    world, rank = ranks.world, ranks.rank
    if args.gpus != world:
        raise SystemExit("--gpus %d does not match WORLD_SIZE %d" % (args.gpus, world))
    if os.environ.get("SLAM_AMD_BENCH_FAIL_RANK") == str(rank):   # test hook: this rank dies after the rendezvous
        os._exit(7)
    n_kf, n_per = min(args.pair_keyframes, 12), min(args.pair_desc, 16)
    pairs = pairs_mod.all_pairs(n_kf)
    dev = torch.device("cpu")
    pg = pairs_mod.PairGather(torch, dist, len(pairs), n_per, world, rank, dev, min(args.pair_chunks, 4),
                              group=ranks.data_plane(dev))

    def rows(lo, hi):
        g = torch.arange(lo, hi, dtype=torch.int64)[:, None] * 1000 + torch.arange(n_per, dtype=torch.int64)[None, :]
        return g.to(torch.int32)

    def compute(a, b):
        pg.res[a:b] = rows(pg.lo + a, pg.lo + b)

    times = timed_repeats(ranks, lambda: pg.run(compute), 1, 2)
    ok = None
    if rank == 0:
        ok = bool(torch.equal(pg.result(), rows(0, len(pairs))))
    ranks.close()
    if rank == 0:
        print(json.dumps({"rehearsal": True, "n_gpus": world, "gather_ok": ok, "pairs": len(pairs),
                          "seconds": round(statistics.median(times), 4)}), flush=True)
    return 0 if ok in (None, True) else 1


def run_stream4k(args):
    """4K stream configuration: 3840x2160 stream, 10000 features, extract + match vs previous frame.  Two legs per
    batch size (frames queued per launch sequence): frames resident in HBM, results left there (`value`); and frames
    arriving in pinned host memory through the library's C++ pipeline (slam_amd_pipeline_*), upload / compute /
    download overlapped on three streams, results read back on the host."""
    import torch
    from slam_amd import capi, synth
    W, H, NF, CAP = 3840, 2160, 10000, 12288
    n = args.stream_frames
    batches = [int(b) for b in args.stream_batches.split(",")]
    seq = synth.frame_sequence(W, H, 8, seed=1)
    params = capi.orb_params(nfeatures=NF)
    dev = torch.device("cuda", 0)
    st = torch.cuda.current_stream().cuda_stream
    legs = []
    for B in batches:
        steps = max(1, n // B)
        # ---- pipelined: the ring's input buffers are written once (a camera's DMA would keep writing there) ----
        pipe = capi.Pipeline(params, W, H, CAP, depth=4, batch=B)
        ring = 4 * B
        for i in range(2 * ring):                                 # fill + warm-up
            buf = pipe.input(i)
            if i < ring:
                buf[...] = seq[i % len(seq)]
            pipe.submit(i)
        pipe.result(2 * ring - 1, copy=False)
        t0 = time.perf_counter()
        for i in range(2 * ring, 2 * ring + steps * B):
            pipe.input(i)                                         # waits until the slot's previous results are out
            pipe.submit(i)
        kp, desc, matches = pipe.result(2 * ring + steps * B - 1)
        dt = time.perf_counter() - t0
        pipe.close()
        # ---- resident: the same launches on frames already in HBM, no result copies ----
        ex = capi.Extractor(params, W, H, max_batch=B, max_keypoints=CAP)
        m = capi.Matcher()
        m.reserve(B, CAP, CAP)
        d_img = [torch.from_numpy(np.stack([seq[(s * B + k) % len(seq)] for k in range(B)])).to(dev) for s in range(2)]
        d_kp = [torch.zeros(B * CAP * 28, dtype=torch.uint8, device=dev) for _ in range(2)]
        d_desc = [torch.zeros(B * CAP * 32, dtype=torch.uint8, device=dev) for _ in range(2)]
        d_cnt = [torch.zeros(B, dtype=torch.int32, device=dev) for _ in range(2)]
        d_out = torch.zeros(B * CAP * 16, dtype=torch.uint8, device=dev)
        d_nout = torch.zeros(B, dtype=torch.int32, device=dev)

        def resident_step(i):
            s_, p_ = i % 2, (i - 1) % 2
            ex.extract_batch_async(d_img[s_], B, H * W, W, d_kp[s_].data_ptr(), d_desc[s_].data_ptr(), d_cnt[s_].data_ptr(), CAP, st)
            m.frame_match_chain_async(B, d_desc[p_].data_ptr() + (B - 1) * CAP * 32, d_kp[p_].data_ptr() + (B - 1) * CAP * 28,
                                      d_cnt[p_].data_ptr() + 4 * (B - 1), d_desc[s_].data_ptr(), d_kp[s_].data_ptr(), CAP,
                                      d_cnt[s_].data_ptr(), d_out, d_nout, max_dist=300.0, area=-1.0, max_level=-1, stream=st)
        for i in range(8):
            resident_step(i)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for i in range(8, 8 + steps):
            resident_step(i)
        torch.cuda.synchronize()
        dt2 = time.perf_counter() - t1
        ex.close(); m.close()
        legs.append({"batch": B, "frames": steps * B, "frames_per_s_resident": round(steps * B / dt2, 2),
                     "frames_per_s_pipelined": round(steps * B / dt, 2),
                     "ms_per_frame_resident": round(dt2 / (steps * B) * 1e3, 4),
                     "ms_per_frame_pipelined": round(dt / (steps * B) * 1e3, 4),
                     "h2d_GBps_pipelined": round(steps * B * W * H / dt / 1e9, 2),
                     "keypoints_last_frame": int(len(kp)), "matches_last_frame": int(len(matches))})
    last = legs[-1]
    print(json.dumps({
        "metric": "frames/s ORB extract+match @3840x2160 10k feats, pipelined stream", "unit": "frames/s",
        "value": last["frames_per_s_resident"], "value_including_h2d_and_d2h": last["frames_per_s_pipelined"],
        "n_gpus": 1, "steps": last["frames"], "by_batch": legs,
        "h2d_bytes_per_frame": W * H, "d2h_bytes_per_frame": CAP * (28 + 32 + 16) + 8,
        "config": {"workload": "3840x2160 synthetic stream, 10000 ORB features, match vs previous frame; `value`: frames "
                               "resident in HBM, %d per launch sequence; pipelined: frames in pinned host memory through "
                               "slam_amd_pipeline_* (C++ host), 3 streams, 4 slots of %d frame(s)" % (last["batch"], last["batch"])},
        "data": "synthetic", "dtype": "u8"}), flush=True)
    return 0


def run_projection(args):
    """KITTI-shaped configuration: 1241x376 frames, 4000 features; 20000 map-point descriptors with
    projected positions matched against the frame's keypoints, window 50 px, maxLevel 4."""
    import torch
    from slam_amd import capi, synth
    W, H, NF, CAP, NQ = 1241, 376, 4000, 8192, 20000
    dev = torch.device("cuda", 0)
    ex = capi.Extractor(capi.orb_params(nfeatures=NF), W, H, max_batch=1, max_keypoints=CAP)
    m = capi.Matcher()
    m.reserve(1, NQ, CAP)
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
    for i in range(max(args.warmup, 4)):
        step(i)
    # eager pass with per-stage events (stage times)
    capi.prof_enable(args.steps * 12 + 16)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(i)
    torch.cuda.synchronize()
    dt_eager = time.perf_counter() - t0
    stages = capi.prof_collect()
    capi.prof_enable(0)
    # timed region: the same step as one HIP graph per input frame, replayed in turn (one frame's six launches
    # are short next to the gaps between eager launches)
    dt, mode = dt_eager, "eager"
    if not args.no_graph:
        try:
            graphs = []
            side = torch.cuda.Stream(device=dev)
            for i in range(4):
                torch.cuda.synchronize()
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, stream=side, capture_error_mode="thread_local"):
                    st = side.cuda_stream
                    step(i)
                graphs.append(g)
            st = torch.cuda.current_stream().cuda_stream
            for i in range(8):
                graphs[i % 4].replay()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for i in range(args.steps):
                graphs[i % 4].replay()
            torch.cuda.synchronize()
            dt, mode = time.perf_counter() - t0, "hip graph replay"
        except Exception as e:
            st = torch.cuda.current_stream().cuda_stream
            sys.stderr.write("graph capture failed (%s); the eager timing is reported\n" % e)
    n_t = int(d_cnt.item())
    print(json.dumps({
        "metric": "frames/s ORB extract + projection-windowed match @1241x376 4k feats vs 20k map points",
        "unit": "frames/s", "value": round(args.steps / dt, 2), "n_gpus": 1, "steps": args.steps,
        "ms_per_step": round(dt / args.steps * 1e3, 4), "launch_mode": mode,
        "ms_per_step_eager_with_stage_events": round(dt_eager / args.steps * 1e3, 4),
        "keypoints": n_t, "matches": int(d_n.item()),
        "match_stage_us": round(sum(stages[k][0] / stages[k][1] for k in ("match", "spatial_sort", "match_window")
                                    if k in stages) * 1e3, 2),
        "equivalent_full_sweep_gcmp_per_s": round(NQ * n_t / sum(
            stages[k][0] / stages[k][1] * 1e-3 for k in ("match", "spatial_sort", "match_window") if k in stages) / 1e9, 1),
        "stage_us_per_launch": {k: round(v[0] / v[1] * 1e3, 2) for k, v in stages.items()},
        "data": "synthetic", "dtype": "u8"}), flush=True)
    ex.close(); m.close()
    return 0


def cpu_baseline(args, synth, seq):
    """The CPU oracle (1 thread) on a bounded sample of the same workload.  Timed on a build of the
    oracle made for this host (gcc -O3 -march=native -ffp-contract=off, so that popcount is one
    instruction), after checking on the first frames that it returns what the -O2 checker returns."""
    from oracle import orb_oracle as oracle
    po = oracle.params(nfeatures=args.nfeatures)
    n = args.cpu_frames
    check = [oracle.detect_and_compute(po, seq[i % len(seq)]) for i in range(2)]
    check_m = oracle.match_gated(check[0][1], check[0][0], check[1][1], check[1][0], 300.0, -1.0, -1)
    build = oracle.use_native_build()
    again = [oracle.detect_and_compute(po, seq[i % len(seq)]) for i in range(2)]
    again_m = oracle.match_gated(again[0][1], again[0][0], again[1][1], again[1][0], 300.0, -1.0, -1)
    same = all(a[0].tobytes() == b[0].tobytes() and np.array_equal(a[1], b[1]) for a, b in zip(check, again)) and \
        check_m.tobytes() == again_m.tobytes()
    if not same:
        raise RuntimeError("the %s build of the oracle disagrees with the -O2 checker" % build)
    prev = oracle.detect_and_compute(po, seq[0])
    t0 = time.perf_counter()
    for i in range(n):
        cur = oracle.detect_and_compute(po, seq[(i + 1) % len(seq)])
        oracle.match_gated(prev[1], prev[0], cur[1], cur[0], 300.0, -1.0, -1)
        prev = cur
    dt = time.perf_counter() - t0
    out = {"value": round(n / dt, 3), "unit": "frames/s", "cores": 1, "kind": "port",
           "sample": "%d frames of the same %dx%d / %d-feature stream: extract + brute-force match vs previous "
                     "frame, CPU oracle (oracle/orb_oracle.c, %s, 1 thread as the reference runs; outputs "
                     "checked equal to the -O2 build on 2 frames)" % (n, args.width, args.height, args.nfeatures, build)}
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
        code = main()
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
    sys.stdout.flush()
    sys.exit(code or 0)

"""Pipelined extract + match over a stream of frames arriving in host memory.

Three HIP streams and a ring of slots: while frame i is extracted and matched
against frame i-1 on the compute stream, frame i+1 is copied host-to-device on the
copy stream and the results of frame i-1 are copied back on a third stream.
Frames are expected in pinned host memory (a camera or decoder would write
there); results land in pinned host buffers of the slot.  PyTorch is used for
pinned memory, streams and events only.
"""
import numpy as np

from . import capi

KP_BYTES, DM_BYTES = 28, 16


class FramePipeline:
    def __init__(self, torch, width, height, params, capacity, depth=3, device=0,
                 max_dist=300.0, area=-1.0, max_level=-1, copy_results=True):
        assert depth >= 3, "the previous frame's slot must outlive the current match"
        self.torch = torch
        self.W, self.H, self.cap, self.depth = width, height, capacity, depth
        self.gates = (max_dist, area, max_level)
        self.copy_results = copy_results
        dev = torch.device("cuda", device)
        self.dev = dev
        self.ex = capi.Extractor(params, width, height, max_batch=1, max_keypoints=capacity, device=device)
        self.m = capi.Matcher(device=device)
        self.m.reserve(1, capacity, capacity)
        self.s_copy = torch.cuda.Stream(device=dev)
        self.s_comp = torch.cuda.Stream(device=dev)
        self.s_back = torch.cuda.Stream(device=dev)
        u8 = torch.uint8
        self.h_img = [torch.empty((height, width), dtype=u8).pin_memory() for _ in range(depth)]
        self.d_img = [torch.empty((height, width), dtype=u8, device=dev) for _ in range(depth)]
        self.d_kp = [torch.zeros(capacity * KP_BYTES, dtype=u8, device=dev) for _ in range(depth)]
        self.d_desc = [torch.zeros(capacity * 32, dtype=u8, device=dev) for _ in range(depth)]
        self.d_cnt = [torch.zeros(1, dtype=torch.int32, device=dev) for _ in range(depth)]
        self.d_match = [torch.zeros(capacity * DM_BYTES, dtype=u8, device=dev) for _ in range(depth)]
        self.d_nmatch = [torch.zeros(1, dtype=torch.int32, device=dev) for _ in range(depth)]
        self.h_kp = [torch.empty(capacity * KP_BYTES, dtype=u8).pin_memory() for _ in range(depth)]
        self.h_desc = [torch.empty(capacity * 32, dtype=u8).pin_memory() for _ in range(depth)]
        self.h_cnt = [torch.zeros(1, dtype=torch.int32).pin_memory() for _ in range(depth)]
        self.h_match = [torch.empty(capacity * DM_BYTES, dtype=u8).pin_memory() for _ in range(depth)]
        self.h_nmatch = [torch.zeros(1, dtype=torch.int32).pin_memory() for _ in range(depth)]
        self.ev_h2d = [torch.cuda.Event() for _ in range(depth)]
        self.ev_done = [torch.cuda.Event() for _ in range(depth)]
        self.ev_out = [torch.cuda.Event() for _ in range(depth)]
        self.frames = 0

    def close(self):
        self.torch.cuda.synchronize()
        self.ex.close()
        self.m.close()

    def slot_image(self, i):
        """Pinned host buffer the producer fills for frame i (waits until the slot is free)."""
        s = i % self.depth
        if i >= self.depth:
            self.ev_out[s].synchronize()
        return self.h_img[s]

    def submit(self, i):
        """Enqueue copy, extract, match and result copy of frame i (its image is in slot_image(i))."""
        torch = self.torch
        s, p = i % self.depth, (i - 1) % self.depth
        with torch.cuda.stream(self.s_copy):
            # the device image of this slot was last read by frame i-depth's extraction
            if i >= self.depth:
                self.s_copy.wait_event(self.ev_done[s])
            self.d_img[s].copy_(self.h_img[s], non_blocking=True)
            self.ev_h2d[s].record(self.s_copy)
        with torch.cuda.stream(self.s_comp):
            self.s_comp.wait_event(self.ev_h2d[s])
            if i >= self.depth:
                self.s_comp.wait_event(self.ev_out[s])   # results of frame i-depth have left the slot
            st = self.s_comp.cuda_stream
            self.ex.extract_batch_async(self.d_img[s], 1, 0, self.W, self.d_kp[s], self.d_desc[s],
                                        self.d_cnt[s], self.cap, st)
            if i > 0:
                md, area, lvl = self.gates
                self.m.frame_match_async(self.d_desc[p], self.d_kp[p], self.cap, self.d_cnt[p],
                                         self.d_desc[s], self.d_kp[s], self.cap, self.d_cnt[s],
                                         self.d_match[s], self.d_nmatch[s], max_dist=md, area=area,
                                         max_level=lvl, stream=st)
            self.ev_done[s].record(self.s_comp)
        with torch.cuda.stream(self.s_back):
            self.s_back.wait_event(self.ev_done[s])
            if self.copy_results:
                self.h_cnt[s].copy_(self.d_cnt[s], non_blocking=True)
                self.h_kp[s].copy_(self.d_kp[s], non_blocking=True)
                self.h_desc[s].copy_(self.d_desc[s], non_blocking=True)
                self.h_nmatch[s].copy_(self.d_nmatch[s], non_blocking=True)
                self.h_match[s].copy_(self.d_match[s], non_blocking=True)
            self.ev_out[s].record(self.s_back)
        self.frames = i + 1

    def result(self, i):
        """Host results of frame i: (keypoints, descriptors, matches vs frame i-1)."""
        s = i % self.depth
        self.ev_out[s].synchronize()
        n = int(self.h_cnt[s][0])
        n = min(n, self.cap)
        kp = self.h_kp[s].numpy().view(capi.KEYPOINT_DTYPE)[:n].copy()
        desc = self.h_desc[s].numpy().reshape(self.cap, 32)[:n].copy()
        nm = int(self.h_nmatch[s][0]) if i > 0 else 0
        matches = self.h_match[s].numpy().view(capi.DMATCH_DTYPE)[:nm].copy()
        return kp, desc, matches

    def run(self, frames, keep_results=False):
        """Feeds an iterable of numpy frames through the pipeline; returns results if asked."""
        out = []
        pending = []
        for i, f in enumerate(frames):
            buf = self.slot_image(i)
            if keep_results and i >= self.depth:
                # the slot about to be reused holds frame i-depth's results
                pass
            buf.numpy()[...] = f
            self.submit(i)
            pending.append(i)
            if keep_results and len(pending) >= self.depth - 1:
                out.append(self.result(pending.pop(0)))
        if keep_results:
            for i in pending:
                out.append(self.result(i))
        self.torch.cuda.synchronize()
        return out

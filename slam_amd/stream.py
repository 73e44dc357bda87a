"""Pipelined extract + match over a stream of frames arriving in host memory.

A thin Python face of the library's C++ pipeline (slam_amd_pipeline_*): three HIP streams and a
ring of slots, so that while frame i is extracted and matched against frame i-1, frame i+1 is
copied to the device and the results of frame i-1 are copied back.  Frames are written into pinned
host buffers the pipeline owns (a camera or decoder would write there); results land in pinned
host buffers of the slot.  No PyTorch here.
"""
from . import capi


class FramePipeline:
    def __init__(self, width, height, params, capacity, depth=3, batch=1, device=0,
                 max_dist=300.0, area=-1.0, max_level=-1):
        self.pipe = capi.Pipeline(params, width, height, capacity, depth=depth, batch=batch, device=device,
                                  max_dist=max_dist, area=area, max_level=max_level)
        self.depth, self.batch = depth, batch
        self.frames = 0

    def close(self):
        self.pipe.close()

    def push(self, image):
        """Copies `image` into the next frame's pinned buffer and submits it; returns the frame's index."""
        i = self.frames
        self.pipe.input(i)[...] = image
        self.pipe.submit(i)
        self.frames = i + 1
        return i

    def result(self, i):
        """Host results of frame i: (keypoints, descriptors, matches vs frame i-1)."""
        return self.pipe.result(i)

    def run(self, frames, keep_results=False):
        """Feeds an iterable of numpy frames through the pipeline; returns every frame's results if asked.
        Results are read `depth - 2` batches behind the newest complete batch, so the copies stay overlapped."""
        out = []
        read = 0
        lag = (self.depth - 2) * self.batch
        for f in frames:
            i = self.push(f)
            done = (i + 1) // self.batch * self.batch       # frames in complete (queued) batches
            while keep_results and read < done - lag:
                out.append(self.result(read)); read += 1
        self.pipe.flush()
        while keep_results and read < self.frames:
            out.append(self.result(read)); read += 1
        if self.frames:
            self.pipe.result(self.frames - 1, copy=False)     # wait for the tail
        return out

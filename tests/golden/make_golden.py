#!/usr/bin/env python3
"""Regenerates the fixtures under tests/golden/ (run in the build container).

chessboard_gray.npz   data/chessboard.png of the reference (its only image asset)
                      converted to 8-bit grey by the oracle's restatement of
                      cv::cvtColor(BGR2GRAY) (main.cpp:124; the 15-bit weights
                      (3735 B + 19235 G + 9798 R + 16384) >> 15 of OpenCV 4.x, the
                      same the library's front end uses).  Needs /root/reference
                      and PIL; the fixture itself is data.
chessboard_orb.npz    outputs of the CPU oracle on that image, resized to width
                      1080 as main.cpp:18,125-129 does (cv::resize INTER_LINEAR,
                      the oracle's orc_resize_linear), and on a homography warp of
                      it, with the parameters of main.cpp:100, plus
                      frameMatch(a, b, {}, 300, -1, 4) as at mapper.cpp:34.

The reference has no golden vectors and OpenCV is not installed, so these pin the
oracle (and through it the GPU path) against regressions; they are not outputs of
the reference itself.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)


def chessboard_pair():
    from oracle import orb_oracle as O
    from slam_amd import synth
    gray = np.load(os.path.join(HERE, "chessboard_gray.npz"))["gray"]
    h, w = gray.shape
    a = O.resize_linear(gray, np.float32(1080) / np.float32(w))      # main.cpp:125-129
    nh, nw = a.shape
    b = synth.warp(a, synth.homography(5.0, 12.0, -7.0, 1.03, nw / 2.0, nh / 2.0))
    return a, b


def main():
    ref_png = "/root/reference/data/chessboard.png"
    out_gray = os.path.join(HERE, "chessboard_gray.npz")
    if os.path.exists(ref_png):
        from PIL import Image
        from oracle import orb_oracle as O
        rgb = np.asarray(Image.open(ref_png).convert("RGB"))
        g = O.bgr2gray(np.ascontiguousarray(rgb[..., ::-1]))           # BGR order, as cv::imread delivers it
        np.savez_compressed(out_gray, gray=g)
    from oracle import orb_oracle as O
    a, b = chessboard_pair()
    p = O.params(nfeatures=1000)
    kp_a, d_a = O.detect_and_compute(p, a)
    kp_b, d_b = O.detect_and_compute(p, b)
    m = O.frame_match(d_a, kp_a, d_b, kp_b, None, 300.0, -1.0, 4)
    np.savez_compressed(os.path.join(HERE, "chessboard_orb.npz"), kp_a=kp_a, desc_a=d_a, kp_b=kp_b,
                        desc_b=d_b, matches=m)
    print("chessboard: %d / %d keypoints, %d matches" % (len(kp_a), len(kp_b), len(m)))


if __name__ == "__main__":
    main()

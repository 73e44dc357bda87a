#!/bin/bash
# Times the reordering kernel (tools/bench_match_frames.py, frame shapes only) with every library build under build_variants/.
for f in build_variants/lib_*.so; do
  echo "== $(basename $f .so)"
  SLAM_AMD_LIB=$PWD/$f timeout -k 10 200 python tools/bench_match_frames.py 0 frames 2>&1 | grep "reordered" || exit 1
done

#!/bin/bash
# Times the matching kernels of every library build under build_variants/ (tools/bench_match_frames.py).
for f in build_variants/lib_*.so; do
  n=$(basename $f .so)
  echo "== $n"
  SLAM_AMD_LIB=$PWD/$f timeout -k 10 150 python tools/bench_match_frames.py 0 ${1:-frames} 2>&1 | grep -v amdgpu.ids || { echo "FAILED $n"; break; }
done

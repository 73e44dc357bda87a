#!/bin/bash
# Times every library build under build_variants/ with the eager per-stage event pass of bench.py
# (experiments with kernel variants; build_variants/ is not tracked).
for f in build_variants/lib_*.so; do
  n=$(basename $f .so)
  SLAM_AMD_LIB=$PWD/$f timeout -k 10 120 python bench.py --no-graph --no-extras --no-pairs --no-cpu --steps 20 --repeats 1 > gpurun_out/var_$n.json 2> gpurun_out/var_$n.err || { echo "FAILED $n"; tail -3 gpurun_out/var_$n.err; break; }
  echo -n "$n "; python tools/show_stages.py gpurun_out/var_$n.json
done

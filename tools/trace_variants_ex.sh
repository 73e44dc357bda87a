#!/bin/bash
# Like trace_variants.sh, but runs only the extraction (variants whose results are wrong cannot run the whole bench).
R=$PWD
cd /tmp && export TMPDIR=/tmp
for f in $R/build_variants/lib_*.so; do
  n=$(basename $f .so)
  export SLAM_AMD_LIB=$f
  timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/tv_$n -o t -- python3 $R/tools/extract_only.py > $R/gpurun_out/tv_$n.txt 2> $R/gpurun_out/tv_$n.err || { echo "FAILED $n"; tail -3 $R/gpurun_out/tv_$n.err; exit 1; }
done

#!/bin/bash
# rocprofv3 kernel + memory-copy trace of a short 4K stream run (tests/cpp/stream4k_host: the C++ pipeline alone), then the copy / compute overlap summary.
# Usage (on the GPU box, from the repository root): bash tools/run_stream_trace.sh <batches> <frames> <tag>
set -e
R=$PWD
OUT=$R/gpurun_out/trace_$3
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 240 rocprofv3 --kernel-trace --memory-copy-trace --stats --output-format csv -d $OUT -o t -- $R/tests/cpp/stream4k_host $1 $2 > $OUT/run.txt 2> $OUT/err.log
cd $R
python tools/copy_overlap.py $OUT $((3840 * 2160 * $1)) > $OUT/overlap.txt
cat $OUT/run.txt $OUT/overlap.txt
ls $OUT | head -20
# the traces themselves are large: keep the summaries
find $OUT -name "*_trace.csv" -size +20M -delete

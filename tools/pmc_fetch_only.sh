#!/bin/bash
# One PMC pass (bytes read at the memory side of L2) over a short bench run.
# Usage (on the GPU box, from the repo root): tools/pmc_fetch_only.sh <tag>
set -u
TAG=${1:-x}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BENCH="python3 $ROOT/bench.py --steps 20 --warmup 3 --no-cpu --no-pairs"
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- $BENCH > $OUT/pmc_fetch_bench.json 2> $OUT/pmc_fetch.err || exit 1
echo done

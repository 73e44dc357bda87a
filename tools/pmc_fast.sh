#!/bin/bash
# One counter pass over bench.py: instruction and LDS counters of the extraction kernels.
# Usage (on the GPU box, from the repo root): tools/pmc_fast.sh <tag>
set -u
TAG=${1:-f}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/pmcf_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BENCH="python3 $ROOT/bench.py --steps 10 --warmup 2 --repeats 1 --no-cpu --no-pairs --no-extras --no-graph"
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU --output-format csv -d $OUT/a -- $BENCH > $OUT/a.json 2> $OUT/a.err || { tail -5 $OUT/a.err; exit 1; }
python3 - $OUT <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
for f in glob.glob(out + "/a/*/*counter_collection.csv"):
    rows = list(csv.DictReader(open(f)))
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in rows:
        agg[r["Kernel_Name"][:40]][r["Counter_Name"]].append((int(r["Grid_Size"]), float(r["Counter_Value"])))
    for k, v in agg.items():
        if not any(s in k for s in ("fast_nms", "describe", "pyramid", "select")):
            continue
        line = []
        for c, vals in sorted(v.items()):
            g = max(x[0] for x in vals)
            big = [x[1] for x in vals if x[0] == g]
            line.append("%s=%.0f" % (c, sum(big) / len(big)))
        print(k, " ".join(line))
PY

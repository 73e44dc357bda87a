#!/bin/bash
# Counter passes over bench.py for the matching kernel (each pass a run of its own).
# Usage (on the GPU box, from the repo root): tools/pmc_match.sh <tag>
set -u
TAG=${1:-mm}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BENCH="python3 $ROOT/bench.py --steps 10 --warmup 2 --repeats 1 --no-cpu --no-pairs --no-extras --no-graph"
timeout -s KILL 120 rocprofv3 --pmc SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT --output-format csv -d $OUT/a -- $BENCH > $OUT/a.json 2> $OUT/a.err || { tail -5 $OUT/a.err; exit 1; }
timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_MFMA SQ_INSTS_SALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM --output-format csv -d $OUT/b -- $BENCH > $OUT/b.json 2> $OUT/b.err || { tail -5 $OUT/b.err; exit 1; }
python3 - $OUT <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
for p in ("a", "b"):
    for f in glob.glob(out + "/" + p + "/*/*counter_collection.csv"):
        rows = list(csv.DictReader(open(f)))
        agg = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in rows:
            agg[r["Kernel_Name"][:48]][r["Counter_Name"]].append((int(r["Grid_Size"]), float(r["Counter_Value"])))
        for k, v in agg.items():
            if "match" not in k and "fast_nms" not in k and "describe" not in k:
                continue
            line = []
            for c, vals in sorted(v.items()):
                g = max(x[0] for x in vals)
                big = [x[1] for x in vals if x[0] == g]
                line.append("%s=%.0f" % (c, sum(big) / len(big)))
            print(k, " ".join(line))
PY

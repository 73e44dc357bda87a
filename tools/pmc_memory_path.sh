#!/bin/bash
# One counter pass over bench.py's step: vector-memory path (address unit, vector L1, requests to L2)
# next to the instruction counts of every kernel.  Usage (GPU box, repo root): bash tools/pmc_memory_path.sh <tag>
set -u
TAG=${1:-x}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/pmc_mem_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BENCH="python3 $ROOT/bench.py --steps 10 --warmup 2 --repeats 1 --no-cpu --no-pairs --no-extras --no-graph"
timeout -s KILL 120 rocprofv3 --pmc TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum SQ_INSTS_VMEM_RD SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_ANY SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU GRBM_GUI_ACTIVE --output-format csv -d $OUT -- $BENCH > $OUT/bench.json 2> $OUT/err.txt || exit 1
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
rows = collections.defaultdict(lambda: collections.defaultdict(float)); calls = collections.Counter()
for f in glob.glob(sys.argv[1] + '/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'].split('(')[0][:40]
        rows[k][r['Counter_Name']] += float(r['Counter_Value'])
        if r['Counter_Name'] == 'SQ_WAVE_CYCLES': calls[k] += 1
for k, v in rows.items():
    n = max(calls[k], 1)
    print(k, 'launches', n, {c: round(x / n / 1e6, 3) for c, x in sorted(v.items())})
PY

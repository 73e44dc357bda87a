#!/bin/bash
# Profiles bench.py on the GPU box: kernel trace + stats, then two PMC passes
# (read and write bytes at the memory side of L2), each in a run of its own.
# Usage (on the GPU box, from the repo root): tools/profile_round.sh <tag>
set -u
TAG=${1:-r01}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BENCH="python3 $ROOT/bench.py --steps 20 --warmup 3 --repeats 1 --no-cpu --no-pairs --no-extras"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $BENCH > $OUT/trace_bench.json 2> $OUT/trace.err || exit 1
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- $BENCH > $OUT/pmc_fetch_bench.json 2> $OUT/pmc_fetch.err || exit 1
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- $BENCH > $OUT/pmc_write_bench.json 2> $OUT/pmc_write.err || exit 1
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VALU --output-format csv -d $OUT/pmc_sq -- $BENCH > $OUT/pmc_sq_bench.json 2> $OUT/pmc_sq.err || exit 1
timeout -s KILL 120 rocprofv3 --pmc SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_ANY --output-format csv -d $OUT/pmc_sq2 -- $BENCH > $OUT/pmc_sq2_bench.json 2> $OUT/pmc_sq2.err || exit 1
timeout -s KILL 120 rocprofv3 --pmc TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_mem -- $BENCH > $OUT/pmc_mem_bench.json 2> $OUT/pmc_mem.err || exit 1
echo done

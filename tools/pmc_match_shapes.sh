#!/bin/bash
# Counter passes over the frame-matching shapes (tools/bench_match_frames.py, frames only): per kernel variant, averages
# over its launches.  Usage (GPU box, repository root): bash tools/pmc_match_shapes.sh
R=$PWD
OUT=$R/gpurun_out/pmc_shapes
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 200 rocprofv3 --pmc SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_MFMA SQ_WAVES GRBM_GUI_ACTIVE --output-format csv -d $OUT/a -o t -- python3 $R/tools/bench_match_frames.py 0 frames > $OUT/a.txt 2> $OUT/a.err || { tail -5 $OUT/a.err; exit 1; }
timeout -s KILL 200 rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_SALU --output-format csv -d $OUT/b -o t -- python3 $R/tools/bench_match_frames.py 0 frames > $OUT/b.txt 2> $OUT/b.err || { tail -5 $OUT/b.err; exit 1; }
python3 - $OUT <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
for p in ("a", "b"):
    for f in glob.glob(out + "/" + p + "/**/*counter_collection.csv", recursive=True):
        agg = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            if "match_mfma" in r["Kernel_Name"] or "match_window_mfma" in r["Kernel_Name"]:
                agg[r["Kernel_Name"].split("sa::")[1][:52]][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, v in sorted(agg.items()):
            print(k, " ".join("%s=%.0f" % (c.replace("SQ_", ""), sum(x) / len(x)) for c, x in sorted(v.items())))
PY

#!/bin/bash
# One counter pass (instruction, LDS and wave-cycle counters of the extraction kernels) over tools/extract_only.py for
# every library build under build_variants/.  Usage (GPU box, repository root): bash tools/pmc_variants.sh
R=$PWD
cd /tmp && export TMPDIR=/tmp
for f in $R/build_variants/lib_*.so; do
  n=$(basename $f .so)
  export SLAM_AMD_LIB=$f
  timeout -s KILL 120 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU --output-format csv -d $R/gpurun_out/pv_$n -o t -- python3 $R/tools/extract_only.py > $R/gpurun_out/pv_$n.txt 2> $R/gpurun_out/pv_$n.err || { echo "FAILED $n"; tail -3 $R/gpurun_out/pv_$n.err; exit 1; }
  echo "== $n"; python3 - $R/gpurun_out/pv_$n <<'PY'
import csv, glob, sys, collections
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        agg[r["Kernel_Name"][:40]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in agg.items():
        if not any(s in k for s in ("fast_nms", "describe", "pyramid", "select")):
            continue
        print("  ", k.replace("sa::", "")[:24], " ".join("%s=%.0f" % (c.replace("SQ_", ""), sum(x) / len(x)) for c, x in sorted(v.items())))
PY
done

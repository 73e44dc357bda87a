#!/bin/bash
# rocprofv3 kernel-trace statistics of bench.py for every library build under build_variants/ (average kernel
# durations without the event overhead of the stage timer).  Usage (GPU box, repository root): bash tools/trace_variants.sh
R=$PWD
cd /tmp && export TMPDIR=/tmp
for f in $R/build_variants/lib_*.so; do
  n=$(basename $f .so)
  export SLAM_AMD_LIB=$f
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/tv_$n -o t -- python3 $R/bench.py --steps 20 --warmup 3 --repeats 1 --no-cpu --no-pairs --no-extras > $R/gpurun_out/tv_$n.json 2> $R/gpurun_out/tv_$n.err || { echo "FAILED $n"; tail -3 $R/gpurun_out/tv_$n.err; exit 1; }
  echo "== $n"; python3 - $R/gpurun_out/tv_$n <<'PY'
import csv, glob, sys, collections, statistics
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    by = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        by[r["Kernel_Name"]].append((int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
    for name, v in by.items():
        if not any(k in name for k in ("fast_nms", "describe", "pyramid", "select_compact", "match_mfma", "finalize")):
            continue
        g = max(x[0] for x in v)
        d = [x[1] / 1e3 for x in v if x[0] == g]
        print("   %-50s %4d launches of the largest grid: mean %8.1f us, median %8.1f us" % (name[:50], len(d), sum(d) / len(d), statistics.median(d)))
PY
done

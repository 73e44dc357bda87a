#!/bin/bash
# Measures the integer ceiling the matching kernels are priced against, on the GPU box:
# builds tools/valu_peak.hip and tools/valu_ops.hip (binaries are not kept in git), runs them,
# keeps their raw stdout under gpurun_out/ and writes the ceiling as JSON.
# Usage (from the repo root, on the GPU box): tools/valu_ceiling.sh <tag>
set -eu
TAG=${1:-r02}
OUT=gpurun_out/valu_$TAG
mkdir -p $OUT
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o $OUT/valu_peak tools/valu_peak.hip
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o $OUT/valu_ops tools/valu_ops.hip
timeout -k 10 120 $OUT/valu_peak > $OUT/valu_peak.txt
timeout -k 10 120 $OUT/valu_ops > $OUT/valu_ops.txt
python3 - "$OUT" <<'PY'
import json, re, sys
out = sys.argv[1]
best = {"sum": 0.0, "key+min": 0.0}
for line in open(out + "/valu_peak.txt"):
    m = re.search(r"chains, (sum|key\+min)\s+blocks=\s*(\d+)\s+([\d.]+) ms\s+([\d.]+) Gcmp/s", line)
    if m:
        best[m.group(1)] = max(best[m.group(1)], float(m.group(4)))
json.dump({"hamming256_gcmp_per_s": best["sum"], "hamming256_with_key_and_min_gcmp_per_s": best["key+min"],
           "source": "tools/valu_peak.hip on one MI355X: best of 1/2/4/8 workgroups per CU; raw output beside this file"},
          open(out + "/valu_peak.json", "w"), indent=1)
print(open(out + "/valu_peak.json").read())
PY
rm -f $OUT/valu_peak $OUT/valu_ops

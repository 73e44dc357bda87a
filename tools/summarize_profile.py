#!/usr/bin/env python3
"""Condenses a tools/profile_round.sh output directory into one text summary for profiles/.
Usage: tools/summarize_profile.py gpurun_out/prof_<tag> [profiles/pmc_counters.json] > profiles/<tag>_rocprofv3_summary.txt"""
import collections
import csv
import glob
import json
import os
import sys

src = sys.argv[1]
out = []


def kernel_short(n):
    n = n.replace("sa::", "").replace("void ", "")
    return n.split("(")[0][:60]


stats = glob.glob(os.path.join(src, "trace", "*", "*kernel_stats.csv"))
if stats:
    out.append("== rocprofv3 --kernel-trace --stats (all calls of the run, warm-up and batch-1 latency loop included) ==")
    out.append("%-62s %7s %12s %12s %12s" % ("kernel", "calls", "avg_us", "min_us", "max_us"))
    for r in csv.DictReader(open(stats[0])):
        out.append("%-62s %7s %12.2f %12.2f %12.2f" % (kernel_short(r["Name"]), r["Calls"], float(r["AverageNs"]) / 1e3,
                                                      float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3))
trace = glob.glob(os.path.join(src, "trace", "*", "*kernel_trace.csv"))
if trace:
    # timed-region launches = the largest grid of each kernel
    rows = list(csv.DictReader(open(trace[0])))
    byk = collections.defaultdict(list)
    for r in rows:
        g = int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"]) if "Grid_Size_X" in r else int(r.get("Grid_Size", 0))
        byk[kernel_short(r["Kernel_Name"])].append((g, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3))
    out.append("")
    out.append("== per kernel, launches with the largest grid only (the full-batch steps of the timed region) ==")
    out.append("%-62s %7s %12s" % ("kernel", "calls", "avg_us"))
    for k, v in sorted(byk.items(), key=lambda kv: -sum(t for g, t in kv[1])):
        gmax = max(g for g, t in v)
        sel = [t for g, t in v if g == gmax]
        out.append("%-62s %7d %12.2f" % (k, len(sel), sum(sel) / len(sel)))
for name in ("pmc_fetch", "pmc_write", "pmc_sq", "pmc_sq2", "pmc_mem"):
    files = glob.glob(os.path.join(src, name, "*", "*counter_collection.csv"))
    if not files:
        continue
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(files[0])):
        agg[kernel_short(r["Kernel_Name"])][r["Counter_Name"]].append((int(r["Grid_Size"]), float(r["Counter_Value"])))
    out.append("")
    out.append("== rocprofv3 --pmc (%s): per-launch value, launches with the largest grid ==" % name)
    for k, cs in agg.items():
        if not k or k.startswith("at::") or k.startswith("__amd"):
            continue
        parts = []
        for c, v in cs.items():
            gmax = max(g for g, _ in v)
            sel = [x for g, x in v if g == gmax]
            parts.append("%s=%.0f" % (c, sum(sel) / len(sel)))
        out.append("%-62s %s" % (k, "  ".join(parts)))
# machine-readable PMC record for bench.py's roofline.traffic
if len(sys.argv) > 2:
    rec = {"kernels": {}}
    vals = collections.defaultdict(dict)
    for name, only in (("pmc_fetch", "FETCH_SIZE"), ("pmc_write", "WRITE_SIZE"), ("pmc_sq", None), ("pmc_sq2", None), ("pmc_mem", None)):
        files = glob.glob(os.path.join(src, name, "*", "*counter_collection.csv"))
        if not files:
            continue
        per = collections.defaultdict(list)
        for r in csv.DictReader(open(files[0])):
            if only and r["Counter_Name"] != only:
                continue
            per[(kernel_short(r["Kernel_Name"]), r["Counter_Name"])].append((int(r["Grid_Size"]), float(r["Counter_Value"])))
        for (k, counter), v in per.items():
            if k.startswith("at::") or k.startswith("__amd") or not k:
                continue
            key = {"FETCH_SIZE": "fetch_size_kib", "WRITE_SIZE": "write_size_kib"}.get(counter, counter.lower())
            gmax = max(g for g, _ in v)
            sel = [x for g, x in v if g == gmax]
            short = k.replace("_kernel", "").split("<")[0]
            if gmax >= vals[short].get("_grid_" + key, 0):   # template variants: keep the largest launch
                vals[short][key] = sum(sel) / len(sel)
                vals[short]["_grid_" + key] = gmax
    rec["kernels"] = {k: {kk: vv for kk, vv in v.items() if not kk.startswith("_")} for k, v in vals.items()
                      if "fetch_size_kib" in v and "write_size_kib" in v}
    try:
        line = json.loads(open(os.path.join(src, "trace_bench.json")).read().strip().splitlines()[-1])
        rec["batch"] = line["config"]["frames_per_step_per_gpu"]
        wl = line["config"]["workload"].split(" ")[0].split("x")
        rec["width"], rec["height"] = int(wl[0]), int(wl[1])
    except Exception:
        pass
    rec["source"] = "rocprofv3 --pmc FETCH_SIZE, --pmc WRITE_SIZE and --pmc SQ_*, separate runs of bench.py; per launch, largest grid"
    json.dump(rec, open(sys.argv[2], "w"), indent=1)
for j in ("trace_bench.json",):
    p = os.path.join(src, j)
    if os.path.exists(p) and os.path.getsize(p):
        out.append("")
        out.append("== bench.py line of the traced run ==")
        out.append(open(p).read().strip())
print("\n".join(out))

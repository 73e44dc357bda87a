import csv, glob, sys, collections, statistics
for d in sorted(glob.glob("gpurun_out/tv_*/")):
    for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
        by = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            by[r["Kernel_Name"]].append((int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
        line = d.split("tv_")[1].strip("/") + ": "
        for name, v in by.items():
            if not any(k in name for k in ("fast_nms", "describe", "pyramid", "select_compact")):
                continue
            g = max(x[0] for x in v)
            dd = [x[1] / 1e3 for x in v if x[0] == g]
            line += "%s %.1f  " % (name.split("(")[0].replace("sa::","").replace("void ","")[:22], statistics.median(dd))
        print(line)

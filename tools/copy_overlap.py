#!/usr/bin/env python3
"""Reads a rocprofv3 run made with --kernel-trace --memory-copy-trace (CSV output) and reports how much of the
host-to-device copy time ran while a kernel was executing, per kernel name: the evidence that the stream pipeline's
uploads overlap its compute.
Only the steady state is looked at: the window that holds the last 60 % of the host-to-device copies (the run's
set-up and warm-up are left out).
Usage: python tools/copy_overlap.py <directory with *_kernel_trace.csv and *_memory_copy_trace.csv> [bytes per upload]"""
import csv
import glob
import os
import sys
from collections import defaultdict


def find(d, suffix):
    hits = glob.glob(os.path.join(d, "**", "*" + suffix), recursive=True)
    if not hits:
        sys.exit("no %s under %s" % (suffix, d))
    return hits[0]


def main():
    d = sys.argv[1]
    upload_bytes = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    kernels = []
    with open(find(d, "kernel_trace.csv")) as f:
        for r in csv.DictReader(f):
            kernels.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0]))
    copies = []
    with open(find(d, "memory_copy_trace.csv")) as f:
        for r in csv.DictReader(f):
            copies.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Direction"], int(r.get("Bytes", 0) or 0)))
    kernels.sort()
    ups = sorted(c for c in copies if c[2].endswith("HOST_TO_DEVICE"))
    t0 = ups[int(len(ups) * 0.4)][0]
    t1 = max(c[1] for c in ups)
    kernels = [(max(s, t0), min(e, t1), n) for s, e, n in kernels if e > t0 and s < t1]
    copies = [(max(s, t0), min(e, t1), dr, (upload_bytes if dr.endswith("HOST_TO_DEVICE") and s >= t0 else 0)) for s, e, dr, _ in copies
              if e > t0 and s < t1]
    print("steady-state window %.3f ms, %d kernel launches, %d copies" % ((t1 - t0) / 1e6, len(kernels), len(copies)))
    busy = sum(e - s for s, e, _ in kernels)
    print("kernels executing: %.3f ms (%.1f %% of the window; launches on one stream, so no double counting)" % (busy / 1e6, 100.0 * busy / (t1 - t0)))
    per = defaultdict(int)
    for s_, e_, n_ in kernels:
        per[n_] += e_ - s_
    for name, ns in sorted(per.items(), key=lambda kv: -kv[1])[:8]:
        print("    %-40s %8.3f ms (%.1f %% of the window)" % (name[:40], ns / 1e6, 100.0 * ns / (t1 - t0)))
    for direction in sorted({c[2] for c in copies}):
        sel = [c for c in copies if c[2] == direction]
        total = sum(e - s for s, e, _, _ in sel)
        nbytes = sum(c[3] for c in sel)
        over = defaultdict(int)
        covered = 0
        ki = 0
        for s, e, _, _ in sorted(sel):
            while ki < len(kernels) and kernels[ki][1] <= s:
                ki += 1
            j = ki
            last = s
            while j < len(kernels) and kernels[j][0] < e:
                ks, ke, name = kernels[j]
                a, b = max(ks, last), min(ke, e)      # (kernels on one stream do not overlap each other; `last` guards the rest)
                if b > a:
                    over[name] += b - a
                    covered += b - a
                    last = b
                j += 1
        rate = ", %.1f GB/s while copying" % (nbytes / max(total, 1)) if nbytes else ""
        print("%s: %d copies, %.3f ms in flight = %.1f %% of the window%s; %.1f %% of the copy time a kernel was executing"
              % (direction, len(sel), total / 1e6, 100.0 * total / (t1 - t0), rate, 100.0 * covered / max(total, 1)))
        for name, ns in sorted(over.items(), key=lambda kv: -kv[1])[:8]:
            print("    %-40s %8.3f ms (%.1f %%)" % (name[:40], ns / 1e6, 100.0 * ns / max(total, 1)))


if __name__ == "__main__":
    main()

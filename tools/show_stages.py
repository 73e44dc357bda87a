#!/usr/bin/env python3
"""Prints the per-stage times and the step time of a bench.py JSON line (file argument)."""
import json
import sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(d["extra"]["stage_us_per_launch"], "ms/step", d["ms_per_step"], "frames/s", d["value"])

#!/usr/bin/env python
"""Average duration of the LAST n launches of a kernel in a rocprofv3 kernel-trace csv (the steady state of a bench run: the
per-kernel --stats summary also averages the ramp-up frames, whose edge lists are shorter).
    python tools/kernel_tail_avg.py trace.csv corr_pyramid [n=20]"""
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if sys.argv[2] in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
n = int(sys.argv[3]) if len(sys.argv) > 3 else 20
d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows]
print(f"{sys.argv[2]}: {len(d)} launches, all: {sum(d) / len(d):.1f} us, last {n}: {sum(d[-n:]) / len(d[-n:]):.1f} us")

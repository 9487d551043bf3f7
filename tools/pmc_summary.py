#!/usr/bin/env python
"""Aggregate a rocprofv3 counter_collection.csv per kernel: mean counter value per dispatch.  usage: pmc_summary.py file.csv [filter]"""
import csv, re, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
flt = sys.argv[2] if len(sys.argv) > 2 else ""
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    name = r.get("Kernel_Name", "")
    if flt and flt not in name: continue
    short = re.sub(r"\(.*", "", re.sub(r"<.*", "", name.replace("(anonymous namespace)::", "")))[:50]
    agg[short][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in agg.items():
    print(k, {c: round(sum(v) / len(v), 1) for c, v in d.items()}, "dispatches", max(len(v) for v in d.values()))

#!/usr/bin/env python
"""Per-frame span, sum of kernel durations and union of busy intervals from a rocprofv3 kernel trace csv."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'normalize_image' in r['Kernel_Name']]
a, b = idx[-22], idx[-2]
seg = rows[a:b]
t0, t1 = int(seg[0]['Start_Timestamp']), int(rows[b]['Start_Timestamp'])
tot = sum(int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in seg)
iv = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp'])) for r in seg)
union = 0; cs, ce = iv[0]
for s, e in iv[1:]:
    if s > ce: union += ce - cs; cs, ce = s, e
    else: ce = max(ce, e)
union += ce - cs
n = 20
print(f"frames {n}: span {(t1-t0)/n/1e3:.1f} us/frame, sum of kernels {tot/n/1e3:.1f}, union busy {union/n/1e3:.1f}, idle {(t1-t0-union)/n/1e3:.1f}")

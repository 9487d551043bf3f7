#!/usr/bin/env python
"""Summarise a rocprofv3 *_kernel_stats.csv with short kernel names.  usage: kstats.py file.csv [n] [filter]"""
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 30
flt = sys.argv[3] if len(sys.argv) > 3 else ""
def short(s):
    s = re.sub(r'\(anonymous namespace\)::', '', s)
    m = re.match(r'_ZN?\d*(?:_GLOBAL__N_1)?(\d+)([A-Za-z_0-9]+)', s)
    if s.startswith('_Z'):
        m = re.search(r'(\d+)([a-z][a-z_0-9]+kernel)', s)
        if m: s = m.group(2) + ('<f32A>' if 'ILb1E' in s else '<f16A>' if 'ILb0E' in s else '')
    s = re.sub(r'^void ', '', s)
    s = re.sub(r'<.*', '', s) if not s.endswith('A>') else s
    s = re.sub(r'\(.*', '', s)
    return s[:60]
tot = sum(float(r['TotalDurationNs']) for r in rows)
print(f"total {tot/1e6:.2f} ms")
k = 0
for r in rows:
    if flt and flt not in r['Name']: continue
    print(f"{short(r['Name']):60s} calls={int(r['Calls']):6d} avg_us={float(r['AverageNs'])/1e3:9.1f} total_ms={float(r['TotalDurationNs'])/1e6:9.2f} {float(r['Percentage']):5.1f}%")
    k += 1
    if k >= n: break

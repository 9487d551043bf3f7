#!/usr/bin/env python
"""Timeline of one steady-state frame from a rocprofv3 kernel trace csv.  usage: frame_trace.py <kernel_trace.csv> [min_gap_us]"""
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
mg = float(sys.argv[2]) if len(sys.argv) > 2 else -1
def short(s):
    m = re.search(r'(\d+)([a-z][a-z_0-9]+kernel)', s)
    if s.startswith('_Z') and m: return m.group(2)
    s = re.sub(r'^void ', '', s); s = re.sub(r'\(anonymous namespace\)::', '', s)
    return re.sub(r'[<(].*', '', s)[:44]
idx = [i for i, r in enumerate(rows) if 'normalize_image' in r['Kernel_Name']]
a, b = idx[-10], idx[-9]
t0 = int(rows[a]['Start_Timestamp']); prev = t0; busy = 0
for r in rows[a:b]:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    g = (s - prev) / 1e3
    if g >= mg: print(f"{(s-t0)/1e3:8.1f} gap={g:6.1f} dur={(e-s)/1e3:6.1f} {short(r['Kernel_Name'])}")
    prev = e; busy += e - s
print("frame span us", (int(rows[b]['Start_Timestamp']) - t0) / 1e3, "busy", busy / 1e3, "launches", b - a)

#!/usr/bin/env python
"""A global-BA frame of BASELINE config 5 from a rocprofv3 kernel-trace csv of tools/lc_profile.py: every kernel from one correlation
launch to the next, for the LAST such stretch that contains the global BA's row kernel; gaps > 5 us on the compute queue are flagged.
    python tools/lc_timeline.py kernel_trace.csv [which_from_the_end=1]"""
import csv
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
back = int(sys.argv[2]) if len(sys.argv) > 2 else 1
corr = [i for i, r in enumerate(rows) if 'corr_pyramid' in r['Kernel_Name']]
frames = [(a, b) for a, b in zip(corr[:-1], corr[1:]) if any('gba_row' in r['Kernel_Name'] for r in rows[a:b])]
a, b = frames[-back]
t0 = int(rows[a]['Start_Timestamp'])
mainq = rows[a]['Queue_Id']
last_end = None
busy = 0.0
for r in rows[a:b]:
    s, e = (int(r['Start_Timestamp']) - t0) / 1e3, (int(r['End_Timestamp']) - t0) / 1e3
    name = re.sub(r'\(anonymous namespace\)::', '', r['Kernel_Name'])
    name = re.sub(r'^void ', '', name)
    gap = ''
    if r['Queue_Id'] == mainq:
        if last_end is not None and s - last_end > 5.0:
            gap = f'   <-- {s - last_end:.0f} us idle before'
        last_end = max(last_end or 0.0, e)
        busy += e - s
    print(f"{s:9.1f} {e:9.1f} {e - s:7.1f}  q{r['Queue_Id']:>2s}  {name[:90]}{gap}")
print(f"frame: {(int(rows[b]['Start_Timestamp']) - t0) / 1e3:.1f} us corr to corr, {b - a} kernels, compute queue busy {busy:.1f} us")

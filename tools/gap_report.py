#!/usr/bin/env python
"""Launch-to-launch gaps on the main queue from a rocprofv3 kernel trace csv (last 20 frames): which kernel pairs the idle time sits between."""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'normalize_image' in r['Kernel_Name']]
a, b = idx[-22], idx[-2]
seg = rows[a:b]
qs = collections.Counter(r['Queue_Id'] for r in seg)
print("queues:", dict(qs))
mainq = qs.most_common(1)[0][0]
m = [r for r in seg if r['Queue_Id'] == mainq]
gaps = collections.defaultdict(list)
tot = 0
for p, n in zip(m[:-1], m[1:]):
    g = int(n['Start_Timestamp']) - int(p['End_Timestamp'])
    gaps[(p['Kernel_Name'][:30], n['Kernel_Name'][:30])].append(g)
    tot += g
ksum = sum(int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in m)
print(f"main queue: {len(m) / 20:.1f} kernels/frame, kernel time {ksum / 20e3:.1f} us/frame, gaps {tot / 20e3:.1f} us/frame")
for k, v in sorted(gaps.items(), key=lambda kv: -sum(kv[1]))[:int(sys.argv[2]) if len(sys.argv) > 2 else 30]:
    print(f"{sum(v) / 20e3:7.2f} us/frame  n/frame={len(v) / 20:4.1f} avg={sum(v) / len(v) / 1e3:6.2f}  {k[0]} -> {k[1]}")

#!/usr/bin/env python
"""One steady-state frame of a rocprofv3 kernel-trace csv as a timeline: start / end (us from the frame's first kernel), queue, kernel.
    python tools/frame_timeline.py kernel_trace.csv [frames_from_the_end=3]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
back = int(sys.argv[2]) if len(sys.argv) > 2 else 3
key = sys.argv[3] if len(sys.argv) > 3 else 'win_hist_kernel'      # once per frame on the compute stream (frame_state_kernel runs twice, win_zero_kernel not at all inside the frame call since round 4)
idx = [i for i, r in enumerate(rows) if key in r['Kernel_Name']]
a, b = idx[-back - 1], idx[-back]
t0 = int(rows[a]['Start_Timestamp'])
for r in rows[a:b]:
    s, e = (int(r['Start_Timestamp']) - t0) / 1e3, (int(r['End_Timestamp']) - t0) / 1e3
    print(f"{s:9.1f} {e:9.1f} {e - s:7.1f}  q{r['Queue_Id']:>2s}  {r['Kernel_Name'][:70]}")

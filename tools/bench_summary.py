#!/usr/bin/env python
"""python tools/bench_summary.py out_dir [runs=2] [bench.py args...]: run bench.py, keep each JSON line, print the headline numbers."""
import json, os, subprocess, sys
out = sys.argv[1]; runs = int(sys.argv[2]) if len(sys.argv) > 2 else 2
os.makedirs(out, exist_ok=True)
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for i in range(runs):
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py")] + sys.argv[3:], capture_output=True, text=True)
    line = r.stdout.strip().split("\n")[-1]
    open(os.path.join(out, f"bench_{i}.json"), "w").write(line + "\n")
    try:
        d = json.loads(line)
        u = d.get("roofline_update") or {}
        print(f"{d['value']:8.1f} frames/sec  {d['ms_per_step']:.4f} ms/frame  update {u.get('avg_ms')} ms {u.get('autotune_ms')}  corr {d['roofline']['avg_launch_ms']} ms")
    except Exception as e:
        print("bench failed:", e, r.stderr[-2000:])

#!/usr/bin/env python
"""Host time between the result record of frame t (event wait returns) and the frame call of frame t + 1 (dpvo_amd.dpvo._HOST_TRACE stamps
inside dpvo_amd/dpvo.py): where the inter-frame gap goes.  Dev tool."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, bench
import dpvo_amd.dpvo as dm
dm._HOST_TRACE = []
from dpvo_amd.config import cfg as base_cfg, DEFAULT_YAML
from dpvo_amd.net import VONet
dev = torch.device("cuda:0")
cfg = base_cfg.clone(); cfg.merge_from_dict(DEFAULT_YAML); cfg.KEYFRAME_THRESH = -1.0
torch.manual_seed(1234)
slam = dm.DPVO(cfg, VONet(), ht=480, wd=640, device=dev, defer_keyframe=True, overlap_encoders=True)
slam.motion_probe = lambda: 1.0e9
frames = bench.make_stream(64, 480, 640, dev)
intr = torch.tensor([320.0, 320.0, 320.0, 240.0], device=dev)
with torch.no_grad():
    for t in range(60):
        slam(float(t), frames[t % 64], intr, image_ready=False)
    del dm._HOST_TRACE[:]
    for t in range(60, 120):
        slam(float(t), frames[t % 64], intr, image_ready=False)
    slam.flush(); torch.cuda.synchronize()
tr = dm._HOST_TRACE
order = ["sync", "fin", "fast", "comp", "fuc", "call", "ret"]
seqs, cur = [], {}
for k, t in tr:
    if k == "sync":
        if cur: seqs.append(cur)
        cur = {}
    cur[k] = t
seqs = [c for c in seqs if all(k in c for k in ("sync", "fin", "fuc", "call", "ret"))]
print(f"{len(seqs)} frames; median host us from the return of the record wait:")
for a, b in (("sync", "fin"), ("fin", "fast"), ("fast", "fuc"), ("fuc", "call"), ("call", "ret")):
    v = [1e6 * (c[b] - c[a]) for c in seqs if a in c and b in c]
    if v: print(f"   {a:5s} -> {b:5s} {np.median(v):7.1f}  (p90 {np.percentile(v, 90):.1f})")
print(f"   sync -> call  {np.median([1e6 * (c['call'] - c['sync']) for c in seqs]):7.1f}")

#!/usr/bin/env python
"""Phases of kf_decide_kernel in a steady-state frame (library built by tools/kf_trace.sh with -DKF_TRACE).  Dev tool."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, bench
from dpvo_amd import _lib as L
from dpvo_amd.config import cfg as base_cfg, DEFAULT_YAML
from dpvo_amd.dpvo import DPVO
from dpvo_amd.net import VONet
dev = torch.device("cuda:0")
cfg = base_cfg.clone(); cfg.merge_from_dict(DEFAULT_YAML); cfg.KEYFRAME_THRESH = -1.0
torch.manual_seed(1234)
slam = DPVO(cfg, VONet(), ht=480, wd=640, device=dev, defer_keyframe=True, overlap_encoders=True)
slam.motion_probe = lambda: 1.0e9
frames = bench.make_stream(64, 480, 640, dev)
intr = torch.tensor([320.0, 320.0, 320.0, 240.0], device=dev)
names = ["start", "flow test done", "classified + block sync", "counts published", "look-back done", "block sync", "indices written",
         "record + host copy", "end"]
rows = []
with torch.no_grad():
    for t in range(100):
        slam(float(t), frames[t % 64], intr, image_ready=False)
        if t >= 90:
            slam.flush(); torch.cuda.synchronize()
            buf = (ctypes.c_ulonglong * 32)()
            assert L.lib().dpvo_debug_kf_trace(buf) == 0
            rows.append(np.array(list(buf), dtype=np.int64).reshape(2, 16)[:, :9])
r = np.median(np.stack(rows), axis=0) / 100.0       # us
t0 = r[0, 0]
for b, nm in ((0, "first chunk"), (1, "last chunk")):
    print(nm + ": " + ", ".join(f"{names[i]} {r[b, i] - t0:+.1f}" for i in range(9)))

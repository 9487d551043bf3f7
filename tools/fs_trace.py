#!/usr/bin/env python
"""Which workgroups make frame_state_kernel take what it takes (library built by tools/fs_trace.sh with -DFS_TRACE).  Dev tool."""
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
M = cfg.PATCHES_PER_FRAME
n_med = (3 * M * 9 + 3) // 4        # kMedPerBlock = 4
with torch.no_grad():
    for t in range(100):
        slam(float(t), frames[t % 64], intr, image_ready=False)
    slam.flush(); torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * (2 * 4096 * 2))()
assert L.lib().dpvo_debug_fs_trace(buf) == 0
a = np.array(list(buf), dtype=np.int64).reshape(2, 4096, 2) / 100.0
for part, name in ((0, "part 1 (state, motion model, median, edges)"), (1, "part 2 (gmap / imap gathers, pyramid level 1)")):
    r = a[part]
    used = r[:, 1] > 0
    t0 = r[used, 0].min()
    print(f"{name}: {used.sum()} workgroups, first start 0.0, last end {r[used, 1].max() - t0:.1f} us")
    def role(b):
        if b < M: return "patches"
        if part == 0:
            if b == M: return "motion model"
            if b < M + 1 + n_med: return "median"
            return "append"
        return "pool" if b > M else "(idle)"
    roles = {}
    for b in np.nonzero(used)[0]:
        roles.setdefault(role(b), []).append((r[b, 0] - t0, r[b, 1] - t0))
    for k, v in roles.items():
        v = np.array(v)
        print(f"   {k:14s} n = {len(v):4d}  start {v[:, 0].min():5.1f} .. {v[:, 0].max():5.1f}   end {v[:, 1].min():5.1f} .. {v[:, 1].max():5.1f}   longest {np.max(v[:, 1] - v[:, 0]):5.1f} us")

#!/usr/bin/env python
"""Which workgroups make ba_patch_kernel take what it takes (library built by tools/ba_trace.sh with -DBA_TRACE).  Dev tool."""
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
with torch.no_grad():
    for t in range(100):
        slam(float(t), frames[t % 64], intr, image_ready=False)
    slam.flush(); torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * 2048)()
assert L.lib().dpvo_debug_ba_trace(buf) == 0
a = np.array(list(buf), dtype=np.int64).reshape(1024, 2)
pb = int(a[1023, 0]); a = a / 100.0
used = np.arange(1023)[a[:1023, 1] > 0]
t0 = a[used, 0].min()
print(f"ba_patch_kernel: {len(used)} workgroups ({pb} per-patch + {len(used) - pb} B-row), last end {a[used, 1].max() - t0:.1f} us")
for name, sel in (("per-patch", used[used < pb]), ("B-row", used[used >= pb])):
    v = a[sel] - t0
    print(f"   {name:10s} n = {len(sel):4d}  start {v[:, 0].min():5.1f} .. {v[:, 0].max():5.1f}   end {v[:, 1].min():5.1f} .. {v[:, 1].max():5.1f}   longest {np.max(v[:, 1] - v[:, 0]):5.1f} us")

sb = (ctypes.c_ulonglong * 40)()
assert L.lib().dpvo_debug_ba_solve_trace(sb) == 0
t = np.array(list(sb), dtype=np.int64) / 100.0
t -= t[0]
print(f"ba_solve60_kernel (thread 0): load + first barrier {t[1]:.1f} us; block steps (panel | barrier | trailing + barrier): " +
      ", ".join(f"{t[2 + 3 * B] - (t[1] if B == 0 else t[4 + 3 * (B - 1)]):.2f}|{t[3 + 3 * B] - t[2 + 3 * B]:.2f}|{t[4 + 3 * B] - t[3 + 3 * B]:.2f}" for B in range(10)) +
      f"; backward substitution {t[32] - t[31]:.1f} us; total {t[32]:.1f} us")

#!/usr/bin/env python
"""Host time of every step of dpvo_frame_update (library built by tools/fu_host_trace.sh with -DFU_HOST_TRACE).  Dev tool."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from dpvo_amd import _lib as L
from dpvo_amd.config import cfg as base_cfg, DEFAULT_YAML
import dpvo_amd.dpvo as _dm
_dm._PLAN_ASIDE = bool(int(os.environ.get("PLAN_ASIDE", "0")))      # (this tool's own switch: the plan on the side stream)
from dpvo_amd.dpvo import DPVO
from dpvo_amd.net import VONet
dev = torch.device("cuda:0")
cfg = base_cfg.clone(); cfg.merge_from_dict(DEFAULT_YAML); cfg.KEYFRAME_THRESH = -1.0
torch.manual_seed(1234)
slam = DPVO(cfg, VONet(), ht=480, wd=640, device=dev, defer_keyframe=True, overlap_encoders=True)
slam.motion_probe = lambda: 1.0e9
frames = bench.make_stream(64, 480, 640, dev)
intr = torch.tensor([320.0, 320.0, 320.0, 240.0], device=dev)
names = ["frame state part 1", "fork record / plan on the compute stream", "reproject", "encoder join + part 2", "correlation",
         "plan stream waits for the fork", "plan launches", "plan-done record", "compute stream waits for the plan", "update operator",
         "update-done records + BA", "keyframe step + record + point cloud"]
out = (ctypes.c_double * 16)()
with torch.no_grad():
    for t in range(60):
        slam(float(t), frames[t % 64], intr, image_ready=False)
    L.lib().dpvo_debug_fu_host_trace(out)
    for t in range(60, 120):
        slam(float(t), frames[t % 64], intr, image_ready=False)
    slam.flush(); torch.cuda.synchronize()
L.lib().dpvo_debug_fu_host_trace(out)
v = list(out)[:12]
print(f"plan on the side stream = {_dm._PLAN_ASIDE}: host us per step of dpvo_frame_update (mean of 60 frames), total {sum(v):.1f}")
for n_, x in zip(names, v):
    print(f"   {n_:44s} {x:7.1f}")

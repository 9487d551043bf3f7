#!/usr/bin/env python
"""Stream-ordered wall-clock stamps of the tracker's two streams without a profiler attached (dpvo_amd.dpvo._STAMPS): per frame, when
the side stream reaches the random draws / the image normalisation / the end of the encoders, and when the main stream reaches
the start / the end of the frame call.  Prints medians relative to the start of the frame call.  Dev tool."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, bench
from dpvo_amd.config import cfg as base_cfg, DEFAULT_YAML
import dpvo_amd.dpvo as _dm
_dm._STAMPS = True
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
    for t in range(140):
        slam(float(t), frames[t % 64], intr, image_ready=False)
    slam.flush(); torch.cuda.synchronize()
s = slam._stamp_buf.cpu().numpy().astype(np.int64) / 100.0          # us
c = slam.counter
rows = [s[(c - 1 - k) % 256] for k in range(2, 60)]
rows = np.array(rows[::-1])
# stamps of frame f: [0] side begin, [1] after rng, [2] after encoders (all for frame f, issued before f's frame call), [3] main before call, [4] after
names = ["side stream reaches frame f's rng", "rng done", "encoders done", "main: frame call begins", "main: frame call done"]
period = np.median(np.diff(rows[:, 3]))
print(f"frame period (main stream, call begin to call begin): median {period:.1f} us")
for i in (0, 1, 2, 4):
    d = rows[:, i] - rows[:, 3]
    print(f"  {names[i]:36s} {np.median(d):+9.1f} us relative to this frame's call begin (p10 {np.percentile(d, 10):+.1f}, p90 {np.percentile(d, 90):+.1f})")
d = rows[1:, 3] - rows[:-1, 4]
print(f"  gap: previous call done -> this call begins   {np.median(d):+9.1f} us")
d = rows[1:, 3] - rows[1:, 2]
print(f"  this call begins after its encoders are done by {np.median(d):+9.1f} us")
d = rows[1:, 0] - rows[:-1, 3]
print(f"  side stream reaches frame f+1's work {np.median(d):+9.1f} us after frame f's call began; its encoders take {np.median(rows[:,2]-rows[:,1]):.1f} us, rng {np.median(rows[:,1]-rows[:,0]):.1f} us")

#!/usr/bin/env python
"""Dev tool: host (Python) time per frame vs GPU time per frame of the bench loop."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
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
    for t in range(45): slam(float(t), frames[t % 64], intr, image_ready=False)
    torch.cuda.synchronize()
    host = []
    t0 = time.perf_counter()
    for t in range(45, 105):
        a = time.perf_counter()
        slam(float(t), frames[t % 64], intr, image_ready=False)
        host.append(time.perf_counter() - a)
    slam.flush(); torch.cuda.synchronize()
    wall = time.perf_counter() - t0
host.sort()
print(f"wall per frame {wall / 60 * 1e3:.3f} ms; host time in slam(): median {host[30] * 1e3:.3f} ms, p10 {host[6] * 1e3:.3f}, p90 {host[54] * 1e3:.3f}")
if len(sys.argv) > 1:
    import cProfile, pstats
    pr = cProfile.Profile(); pr.enable()
    with torch.no_grad():
        for t in range(105, 165): slam(float(t), frames[t % 64], intr, image_ready=False)
    pr.disable(); slam.flush(); torch.cuda.synchronize()
    pstats.Stats(pr).sort_stats(sys.argv[1] if sys.argv[1] in ("tottime", "cumulative") else "cumulative").print_stats(45)

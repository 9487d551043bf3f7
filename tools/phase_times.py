#!/usr/bin/env python
"""Where a steady-state frame's time goes on the MAIN stream, measured with HIP events in an un-profiled run (the profiler
slows the host down and changes the picture): events are recorded at phase boundaries by wrapping a few methods.
  bubble  = end of frame t's last launch (flow test + read-back) -> first launch of frame t+1 (edge removal): the GPU's main
            queue is empty here while the host waits for the keyframe decision
Dev tool."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from dpvo_amd.config import cfg as base_cfg, DEFAULT_YAML
from dpvo_amd.dpvo import DPVO
from dpvo_amd.net import VONet

dev = torch.device("cuda:0")
cfg = base_cfg.clone(); cfg.merge_from_dict(DEFAULT_YAML); cfg.KEYFRAME_THRESH = -1.0
torch.manual_seed(1234)
slam = DPVO(cfg, VONet(), ht=480, wd=640, device=dev, defer_keyframe=True, overlap_encoders=bool(int(os.environ.get("OVERLAP_ENC", "1"))))       # (this tool's own switch)
slam.motion_probe = lambda: 1.0e9
frames = bench.make_stream(64, 480, 640, dev)
intr = torch.tensor([320.0, 320.0, 320.0, 240.0], device=dev)
marks = []          # (name, event) in stream order


import time
def mark(name):
    ev = torch.cuda.Event(enable_timing=True)
    ev.record()
    marks.append((name, ev, time.perf_counter()))


def wrap(obj, attr, before=None, after=None):
    f = getattr(obj, attr)

    def g(*a, **k):
        if before: mark(before)
        r = f(*a, **k)
        if after: mark(after)
        return r
    setattr(obj, attr, g)


from dpvo_amd import projective_ops as pops
wrap(slam, "remove_factors", before="first_launch", after="removed")
wrap(slam, "append_frame_factors", before="frame_state_done", after="appended")
wrap(slam, "reproject", before="plan_done")
wrap(pops, "point_cloud", after="points_done")
wrap(pops, "motionmag_pair", after="flowtest_launched")
wrap(slam, "plan", before="plan_begin")
wrap(slam, "corr", before="corr_begin", after="corr_end")
wrap(slam.network.update, "forward", after="update_end")
wrap(slam, "_keyframe_begin", before="ba_end", after="frame_end")
with torch.no_grad():
    for t in range(60): slam(float(t), frames[t % 64], intr, image_ready=False)
    marks.clear()
    for t in range(60, 160): slam(float(t), frames[t % 64], intr, image_ready=False)
    slam.flush()
torch.cuda.synchronize()
import collections
acc = collections.defaultdict(list)
hacc = collections.defaultdict(list)
for (n0, e0, h0), (n1, e1, h1) in zip(marks[:-1], marks[1:]):
    acc[f"{n0} -> {n1}"].append(e0.elapsed_time(e1)); hacc[f"{n0} -> {n1}"].append(h1 - h0)
tot = 0
for k, v in acc.items():
    if len(v) < 50: continue
    v.sort(); m = sum(v) / len(v); tot += m
    hv = sorted(hacc[k])
    print(f"{k:32s} gpu mean {m * 1e3:7.1f} us   median {v[len(v) // 2] * 1e3:7.1f}   p90 {v[int(len(v) * .9)] * 1e3:7.1f}   | host median {hv[len(hv) // 2] * 1e6:7.1f} us")
print(f"sum {tot * 1e3:.1f} us per frame")

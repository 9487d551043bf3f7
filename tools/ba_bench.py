#!/usr/bin/env python
"""fastba.BA (2 iterations, E = 45 312, 10 free poses) alone: HIP-event time per call and a checksum of the result (bit-identity across
library variants: DPVO_HIP_LIB).  Dev tool."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dpvo_amd import fastba, synthetic as S
from dpvo_amd import projective_ops as pops
from dpvo_amd.graph import GraphPlan
dev = torch.device("cuda:0")
ii, jj, kk = (t.to(dev) for t in S.replay_graph(40))
E = ii.numel()
poses, patches, intr = (t.to(dev) for t in S.make_scene(40))
plan = GraphPlan(ii, jj, kk, n_frames=4096, n_patch_ids=4096 * 96)
torch.manual_seed(0)
coords = pops.transform_coords(poses, patches, intr, ii, jj, kk)
target = coords[0, :, :, 1, 1].contiguous() + 0.5 * torch.randn(E, 2, device=dev)
weight = torch.rand(E, 2, device=dev)
p0, pt0 = poses.clone(), patches.clone()
def run():
    poses.copy_(p0); patches.copy_(pt0)
    fastba.BA(poses, patches, intr, target, weight, 1e-4, ii, jj, kk, 30, 40, M=96, iterations=2, plan=plan)
for _ in range(5): run()
torch.cuda.synchronize()
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
reps = 50
s.record()
for _ in range(reps): run()
e.record(); torch.cuda.synchronize()
chk = int(torch.cat([poses.flatten(), patches.flatten()]).view(torch.int32).long().sum())
print(f"BA x2 iterations (+ 2 state copies): {s.elapsed_time(e) / reps * 1e3:.1f} us per call; result checksum {chk:x}")

#!/usr/bin/env python
"""Dev experiment: plan build time vs key width (rocPRIM radix passes)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dpvo_amd import synthetic as S
from dpvo_amd.graph import GraphPlan
dev = torch.device("cuda:0")
ii, jj, kk = (t.to(dev) for t in S.replay_graph(40))
f0 = int(min(ii.min(), jj.min())); k0 = int(kk.min())
iir, jjr, kkr = ii - f0, jj - f0, kk - k0
def t(fn, reps=30):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / reps * 1e3
print("generic 64-bit        %.1f us" % t(lambda: GraphPlan(ii, jj, kk, n_patches_ub=3000, n_pairs_ub=800)))
print("ranged 4096 / 393216  %.1f us" % t(lambda: GraphPlan(ii, jj, kk, n_patches_ub=3000, n_pairs_ub=800, n_frames=4096, n_patch_ids=4096 * 96)))
nf = int(max(iir.max(), jjr.max())) + 1; npid = int(kkr.max()) + 1
print("relative %d / %d     %.1f us" % (nf, npid, t(lambda: GraphPlan(iir, jjr, kkr, n_patches_ub=3000, n_pairs_ub=800, n_frames=nf, n_patch_ids=npid))))
print("relative 16 bits cap   %.1f us" % t(lambda: GraphPlan(iir, jjr, kkr, n_patches_ub=3000, n_pairs_ub=800, n_frames=16, n_patch_ids=4096)))

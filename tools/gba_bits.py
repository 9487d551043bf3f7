#!/usr/bin/env python
"""sha1 of the poses and depths a global BA call (2 iterations) leaves behind, and of the system (S, y) its first linearisation forms,
on the synthetic problems of tools/gba_bench.py -- run under two builds of the library (DPVO_HIP_LIB) to show a kernel change is bit
neutral:   python tools/gba_bits.py [sizes=50,100,130,200]        (tools/gba_bv_ab.sh runs it for the row kernel's B / v part)"""
import hashlib
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dpvo_amd import fastba, synthetic as S            # noqa: E402
from dpvo_amd.fastba import global_ba as G             # noqa: E402
from dpvo_amd.graph import GraphPlan                   # noqa: E402
from dpvo_amd import projective_ops as pops            # noqa: E402
from dpvo_amd import workspace                         # noqa: E402

dev = torch.device("cuda:0")
M = 96
h = lambda t: hashlib.sha1(t.detach().cpu().contiguous().numpy().tobytes()).hexdigest()[:16]
for n in (int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "50,100,130,200").split(",")):
    cfg = S.GraphCfg(M=M, REMOVAL_WINDOW=10 * n, PATCH_LIFETIME=13)
    ii, jj, kk = S.replay_graph(n, cfg)
    old = torch.arange(3, n - 40, max(1, (n - 43) // 12))[:12]
    ks = (old[:, None] * M + torch.arange(M)[None]).reshape(-1).repeat_interleave(3)
    js = torch.stack([n - 20 + (old % 7), n - 12 + (old % 5), n - 6 + (old % 3)], 1).repeat_interleave(M, 0).reshape(-1)
    ii = torch.cat([ii, ks // M]); jj = torch.cat([jj, js]); kk = torch.cat([kk, ks])
    poses, patches, intr = S.make_scene(n, M=M, seed=1)
    d = lambda t: t.to(dev)
    ii, jj, kk, poses, patches, intr = d(ii), d(jj), d(kk), d(poses), d(patches), d(intr)
    coords = pops.transform_coords(poses, patches, intr, ii, jj, kk)
    g = torch.Generator(device="cpu").manual_seed(0)
    target = coords[0, :, :, 1, 1].contiguous() + 0.5 * torch.randn(ii.numel(), 2, generator=g).to(dev)
    weight = torch.rand(ii.numel(), 2, generator=g).to(dev)
    plan = GraphPlan(ii, jj, kk)
    t0 = 1 if n % 2 else 7          # (free poses from 1 resp. 7: sources outside the free range too)
    fastba.BA(poses, patches, intr, target, weight, 1e-4, ii, jj, kk, t0, n, M=M, iterations=1, eff_impl=True, plan=plan)
    torch.cuda.synchronize()
    n6 = 6 * (n - t0)
    sy = workspace.get(0, dev, "gba_sys").view(torch.float32)          # (the damped system's scratch buffer: S and y of the call above)
    print(f"N = {n - t0:4d}  E = {ii.numel():7d}  S {h(sy[:n6 * n6])}  y {h(sy[n6 * n6:n6 * n6 + n6])}  poses {h(poses)}  depths {h(patches)}", flush=True)
    fastba.BA(poses, patches, intr, target, weight, 1e-4, ii, jj, kk, t0, n, M=M, iterations=2, eff_impl=True, plan=plan)
    torch.cuda.synchronize()
    print(f"          after two more iterations: poses {h(poses)}  depths {h(patches)}  finite {bool(torch.isfinite(poses).all())}", flush=True)

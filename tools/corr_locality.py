#!/usr/bin/env python
"""Dev experiment: memory-side traffic and time of corr_pyramid_kernel under different edge orders.
Run under `rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv` (see tools/pmc.sh); launches are grouped 6 per variant."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dpvo_amd import altcorr
from dpvo_amd import synthetic as S
from dpvo_amd import projective_ops as pops
dev = torch.device("cuda:0")
ii, jj, kk = (t.to(dev) for t in S.replay_graph(40))
E = ii.numel()
gmap, f0, f1, imap = S.make_features()
g = gmap.permute(0, 2, 3, 1).reshape(-1, 9, 128).contiguous().to(dev)
a = f0.permute(0, 2, 3, 1).contiguous().to(dev); b = f1.permute(0, 2, 3, 1).contiguous().to(dev)
poses, patches, intr = (t.to(dev) for t in S.make_scene(40))
us, vs = kk % 3456, jj % 36
coords = pops.transform_coords(poses, patches, intr, ii, jj, kk)
variants = {"edge order": None, "by target frame": torch.argsort(vs, stable=True).int()}
cy = coords.reshape(E, 2, 3, 3)[:, 1, 1, 1]
cy = torch.nan_to_num(cy, nan=0.0, posinf=0.0, neginf=0.0)
key = vs * 4096 + (cy.clamp(0, 119) / 8).long()
variants["by target frame, 8-row band"] = torch.argsort(key, stable=True).int()
for name, order in variants.items():
    for _ in range(3): altcorr.corr_pyramid(g, a, b, coords, us, vs, order=order)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(3): altcorr.corr_pyramid(g, a, b, coords, us, vs, order=order)
    e.record(); torch.cuda.synchronize()
    print(f"{name:32s} {s.elapsed_time(e) / 3 * 1e3:8.1f} us")

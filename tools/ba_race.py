#!/usr/bin/env python
"""Which BA intermediate differs between repetitions when another stream keeps the GPU busy: one Gauss-Newton iteration,
the whole "ba" workspace diffed against the first repetition.  Dev tool."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dpvo_amd import fastba, synthetic as S, workspace
from dpvo_amd import projective_ops as pops
from dpvo_amd.encoders import HipEncoders
from dpvo_amd.graph import GraphPlan
from dpvo_amd.net import VONet

dev = torch.device("cuda:0")
torch.manual_seed(0)
vo = VONet().to(dev)
enc = HipEncoders(vo.patchify.fnet, vo.patchify.inet)
img = (torch.randn(3, 480, 640, device=dev) / 2).half()
side = torch.cuda.Stream(device=dev)
ii, jj, kk = (t.to(dev) for t in S.replay_graph(40))
E = ii.numel()
poses, patches, intr = (t.to(dev) for t in S.make_scene(40))
plan0 = GraphPlan(ii, jj, kk, n_frames=4096, n_patch_ids=4096 * 96)
coords = pops.transform_coords(poses, patches, intr, ii, jj, kk)
target = coords[0, :, :, 1, 1].contiguous() + 0.5 * torch.randn(E, 2, device=dev)
weight = torch.rand(E, 2, device=dev)
p0, pt0 = poses.clone(), patches.clone()
N = 10
al = lambda x: (x + 255) & ~255
PAIR, EDGE, SENT, CHUNK, MAXD = [int(x) for x in sys.argv[1:6]]
n6 = 6 * N
nblk = (E + CHUNK - 1) // CHUNK
names, offs = ["pairbuf", "edgebuf", "Qbuf", "ubuf", "Ecol", "spart", "Sg", "yg", "dX", "end"], [0]
for sz in (E * PAIR * 4, E * EDGE * 4, E * 4, E * 4, E * n6 * 4, nblk * SENT * 4, MAXD * MAXD * 4, MAXD * 4, MAXD * 4):
    offs.append(offs[-1] + al(sz))
ref = None
shown = 0
iters = int(os.environ.get("ITERS", "1"))
for r in range(60):
    if r % 3 == 0:
        with torch.cuda.stream(side):
            for _ in range(6):
                enc(img)
    poses.copy_(p0); patches.copy_(pt0)
    fastba.BA(poses, patches, intr, target, weight, 1e-4, ii, jj, kk, 30, 40, M=96, iterations=iters, plan=plan0)
    ws = workspace.get(1, dev, "ba")[:offs[-1]].clone()
    if ref is None:
        ref = ws; refp = poses.clone()
        continue
    d = (ws != ref).nonzero().flatten()
    if d.numel():
        d = d.cpu()
        hits = {}
        for k in range(len(names) - 1):
            m = ((d >= offs[k]) & (d < offs[k + 1]))
            if m.any():
                x = d[m]
                hits[names[k]] = (int(m.sum()), int(x.min() - offs[k]) // 4, int(x.max() - offs[k]) // 4)
        print("rep", r, "poses equal:", bool(torch.equal(poses, refp)), hits)
        if shown < 2:
            shown += 1
            a = ws[offs[1]:offs[2]].view(torch.float32).view(-1, EDGE); b = ref[offs[1]:offs[2]].view(torch.float32).view(-1, EDGE)
            rows = (a != b).any(1).nonzero().flatten()
            print("  edgebuf rows differing:", rows.numel(), "first rows:", rows[:12].tolist())
            pp = plan0.perm_p.long(); pos = torch.empty_like(pp); pos[pp] = torch.arange(pp.numel(), device=dev)
            print("  their positions in pair order:", pos[rows[:12]].tolist(), " pair ids:", plan0.pu[rows[:12]].tolist())
            pr_ = plan0.pu[rows].long(); lp = pos[rows] - plan0.pair_off.long()[pr_]
            for gq in pr_.unique().tolist():
                m = pr_ == gq
                cols = (a[rows[m]] != b[rows[m]]).any(0).nonzero().flatten().tolist()
                print("   pair", gq, "size", int(plan0.pair_off[gq + 1] - plan0.pair_off[gq]), "bad local positions", sorted(lp[m].tolist()), "fields", cols)
            for e in rows[:3].tolist():
                print("   e", e, "now", [f"{x:.6g}" for x in a[e].tolist()], "\n        ref", [f"{x:.6g}" for x in b[e].tolist()], "\n        now-ref", [f"{x:.3g}" for x in (a[e] - b[e]).tolist()])
            A2 = ws[offs[0]:offs[1]].view(torch.float32).view(-1, PAIR); B2 = ref[offs[0]:offs[1]].view(torch.float32).view(-1, PAIR)
            pr = (A2 != B2).any(1).nonzero().flatten()
            print("  pairbuf rows (pairs) differing:", pr.tolist()[:20])
            if pr.numel():
                g = int(pr[0]); dd = (A2[g] != B2[g]).nonzero().flatten()
                print("   pair", g, "ij", plan0.pair_ij.view(-1, 2)[g].tolist(), "n diff entries", dd.numel(), "max rel", float(((A2[g] - B2[g]).abs() / (B2[g].abs() + 1e-20)).max()))
                print("   Gram (row, col) differing:", [(int(x) // 16, int(x) % 16) for x in dd.tolist()])
                print("   Gram now-ref on those:", [f"{float(A2[g][x] - B2[g][x]):.4g}" for x in dd.tolist()])
print("done")

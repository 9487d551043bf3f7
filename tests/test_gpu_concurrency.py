"""A kernel's result must not depend on what else is resident on the GPU.  Every hot-path op is repeated on fixed inputs
while a second HIP stream runs encoder forward passes (what `overlap_encoders` does in the pipeline); all repetitions must be
bit-identical.  (BA has its own, longer test in test_gpu_ba.py: it is the op that exposed the packed-FP32 hazard the library
is now built around, csrc/Makefile NOPK.)"""
import pytest
import torch

from dpvo_amd import altcorr, synthetic as S
from dpvo_amd import projective_ops as pops
from dpvo_amd.encoders import HipEncoders
from dpvo_amd.graph import GraphPlan
from dpvo_amd.net import VONet

pytestmark = pytest.mark.gpu


def _bits(t):
    t = t.contiguous()
    v = t.view(torch.int16) if t.element_size() == 2 else (t.view(torch.int32) if t.element_size() == 4 else t)
    return v.long().sum()


def test_ops_repeatable_while_another_stream_is_busy(dev):
    torch.manual_seed(0)
    vo = VONet().to(dev)
    enc = HipEncoders(vo.patchify.fnet, vo.patchify.inet)
    img = (torch.randn(3, 480, 640, device=dev) / 2).half()
    eo = (torch.empty(120, 160, 128, dtype=torch.float16, device=dev), torch.empty(120, 160, 384, dtype=torch.float16, device=dev))
    side = torch.cuda.Stream(device=dev)
    ii, jj, kk = (t.to(dev) for t in S.replay_graph(12, S.GraphCfg(REMOVAL_WINDOW=1000)))     # 13 824 edges: the WS GEMM path
    E = ii.numel()
    gmap, f0, f1, imap = S.make_features()
    g = gmap.permute(0, 2, 3, 1).reshape(-1, 9, 128).contiguous().to(dev)
    a = f0.permute(0, 2, 3, 1).contiguous().to(dev); b = f1.permute(0, 2, 3, 1).contiguous().to(dev)
    poses, patches, intr = (t.to(dev) for t in S.make_scene(40))
    imap = imap.to(dev)
    us, vs = kk % 3456, jj % 36
    plan = GraphPlan(ii, jj, kk)
    coords = pops.transform_coords(poses, patches, intr, ii, jj, kk)
    corr = altcorr.corr_pyramid(g, a, b, coords, us, vs)
    net0 = torch.randn(1, E, 384, device=dev)

    def op_update():
        n, (d, w, _) = vo.update(net0.clone(), imap[None], corr[None], None, ii, jj, kk, plan=plan, inp_rows=kk, inp_mod=3456,
                                 corr_is_padded=True)
        return [n, d, w]

    ops = {"plan": lambda: [p.perm_k, p.ku, p.ix, p.jx, p.perm_p, p.pu] if (p := GraphPlan(ii, jj, kk)) else None,
           "reproject": lambda: [pops.transform_coords(poses, patches, intr, ii, jj, kk)],
           "corr": lambda: [altcorr.corr_pyramid(g, a, b, coords, us, vs)],
           "update": op_update}
    reps = 30
    for name, op in ops.items():
        out = None
        for r in range(reps):
            if r % 3 == 0:
                with torch.cuda.stream(side):
                    for _ in range(4):
                        enc(img, fmap_out=eo[0], imap_out=eo[1])
            res = op()
            if out is None:
                out = torch.zeros(reps, len(res), dtype=torch.int64, device=dev)
            for c, t in enumerate(res):
                out[r, c] = _bits(t)
        torch.cuda.synchronize()
        o = out.cpu()
        bad = int((o != o[0:1]).any(1).sum())
        assert bad == 0, f"{name}: {bad} of {reps} repetitions differ while another stream is busy"

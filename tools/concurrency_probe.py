#!/usr/bin/env python
"""Bit-reproducibility of every hot-path op WHILE another stream keeps the GPU busy (encoder forward passes in a loop):
each op runs `reps` times on the same inputs on the main stream, its outputs are bit-checksummed on the device, and the
checksums must all be equal.  Finds kernels whose result depends on what else is resident (uninitialised LDS / register
reads, missing barriers).  Dev tool:  python tools/concurrency_probe.py [n_frames_for_graph ...]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dpvo_amd import altcorr, fastba, synthetic as S          # noqa: E402
from dpvo_amd import net as N                                  # noqa: E402
from dpvo_amd import projective_ops as pops                    # noqa: E402
from dpvo_amd.encoders import HipEncoders                      # noqa: E402
from dpvo_amd.graph import GraphPlan                           # noqa: E402
from dpvo_amd.net import VONet                                 # noqa: E402


def bits(t):
    t = t.contiguous()
    v = t.view(torch.int16) if t.element_size() == 2 else (t.view(torch.int32) if t.element_size() == 4 else t)
    return v.long().sum()


def main():
    dev = torch.device("cuda:0")
    reps = int(os.environ.get('REPS', '24'))
    torch.manual_seed(0)
    vo = VONet().to(dev)
    enc = HipEncoders(vo.patchify.fnet, vo.patchify.inet)
    img = (torch.randn(3, 480, 640, device=dev) / 2).half()
    side = torch.cuda.Stream(device=dev)

    mmA = torch.randn(4096, 4096, device=dev, dtype=torch.float16); mmB = torch.randn(4096, 4096, device=dev, dtype=torch.float16)
    mmC = torch.empty(4096, 4096, device=dev, dtype=torch.float16)
    enc_out = (torch.empty(120, 160, 128, dtype=torch.float16, device=dev), torch.empty(120, 160, 384, dtype=torch.float16, device=dev))

    def busy(k=6, what="enc"):
        with torch.cuda.stream(side):
            for _ in range(k):
                if what == "enc":
                    enc(img, fmap_out=enc_out[0], imap_out=enc_out[1])
                else:
                    torch.matmul(mmA, mmB, out=mmC)

    upd = vo.update
    for nfr in [int(a) for a in sys.argv[1:]] or [8, 40]:
        cfgM = 96
        if nfr == 40:
            ii, jj, kk = (t.to(dev) for t in S.replay_graph(40))
        else:                                          # the initialisation graph: frames 0..nfr-1, nothing removed yet
            ii, jj, kk = (t.to(dev) for t in S.replay_graph(nfr, S.GraphCfg(M=cfgM, REMOVAL_WINDOW=1000)))
        E = ii.numel()
        gmap, f0, f1, imap = S.make_features()
        g = gmap.permute(0, 2, 3, 1).reshape(-1, 9, 128).contiguous().to(dev)
        a = f0.permute(0, 2, 3, 1).contiguous().to(dev); b = f1.permute(0, 2, 3, 1).contiguous().to(dev)
        poses, patches, intr = (t.to(dev) for t in S.make_scene(40))
        imap = imap.to(dev)
        us, vs = kk % 3456, jj % 36
        plan0 = GraphPlan(ii, jj, kk, n_frames=4096, n_patch_ids=4096 * 96)
        coords = pops.transform_coords(poses, patches, intr, ii, jj, kk)
        corr = altcorr.corr_pyramid(g, a, b, coords, us, vs)
        net0 = torch.randn(1, E, 384, device=dev)
        target = coords[0, :, :, 1, 1].contiguous() + 0.5 * torch.randn(E, 2, device=dev)
        weight = torch.rand(E, 2, device=dev)
        p0, pt0 = poses.clone(), patches.clone()
        t0 = 1 if nfr < 40 else 30
        t1 = nfr

        def op_plan():
            p = GraphPlan(ii, jj, kk, n_frames=4096, n_patch_ids=4096 * 96)
            return [p.perm_k, p.ku, p.ix, p.jx, p.perm_p, p.pu, p.counts[:2]]

        def op_reproject():
            return [pops.transform_coords(poses, patches, intr, ii, jj, kk)]

        def op_corr():
            return [altcorr.corr_pyramid(g, a, b, coords, us, vs)]

        def op_update():
            n, (d, w, _) = upd(net0.clone(), imap[None], corr[None], None, ii, jj, kk, plan=plan0, inp_rows=kk, inp_mod=3456,
                               corr_is_padded=True)
            return [n, d, w]

        def op_ba():
            poses.copy_(p0); patches.copy_(pt0)
            fastba.BA(poses, patches, intr, target, weight, 1e-4, ii, jj, kk, t0, t1, M=96, iterations=2, plan=plan0)
            return [poses, patches]

        ops = (("plan", op_plan), ("reproject", op_reproject), ("corr", op_corr), ("update", op_update), ("ba", op_ba))
        if os.environ.get("ONLY"):
            ops = [o for o in ops if o[0] in os.environ["ONLY"].split(",")]
        for name, op in ops:
            for mode in ("alone", "concurrent", "concurrent_mm"):
                out = None
                torch.cuda.synchronize()
                for r in range(reps):
                    if mode != "alone" and r % 3 == 0:
                        busy(what="enc" if mode == "concurrent" else "mm")
                    res = op()
                    if out is None:
                        out = torch.zeros(reps, len(res), dtype=torch.int64, device=dev)
                    for c, t in enumerate(res):
                        out[r, c] = bits(t)
                torch.cuda.synchronize()
                o = out.cpu()
                bad = (o != o[0:1]).any(1).sum().item()
                print(f"E={E:6d} {name:10s} {mode:10s} {'OK' if bad == 0 else 'MISMATCH in %d of %d reps' % (bad, reps)}")


if __name__ == "__main__":
    main()

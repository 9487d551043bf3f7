#!/usr/bin/env python
"""Dev aid (this container, no GPU): runs oracle/ref_pipeline.py's REFERENCE tracker on the CPU for a few frames to shake out the
Python plumbing before a GPU call is spent on it.  device='cuda' factories are mapped to the CPU and cuda_corr / cuda_ba are served
by the C oracle (tests/golden/make_golden_graph.py does the same for the bookkeeping goldens).  Not a test, not a measurement."""
import os
import sys
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import oracle                                   # noqa: E402
from make_golden_graph import cuda_to_cpu      # noqa: E402
from oracle import ref_pipeline as RP          # noqa: E402


def cpu_native():
    cc = types.ModuleType("cuda_corr")

    def forward(fmap1, fmap2, coords, ii, jj, radius):
        out = oracle.corr_forward(fmap1[0].float().numpy(), fmap2[0].float().numpy(), coords[0].float().numpy(), ii.numpy(), jj.numpy(),
                                  radius, np.float32)                    # [E, D, D, P, P] (y, x) like the kernel's raw volume
        o = torch.from_numpy(np.ascontiguousarray(out))[None].to(fmap1.dtype)
        # the extension returns the bilinear-blended, permuted volume (correlation_kernel.cu:221-232); the oracle entry does too
        return [o]

    def patchify_forward(net, coords, radius):
        D = 2 * radius + 2
        B, M = coords.shape[:2]
        out = torch.zeros(B, M, net.shape[1], D, D, dtype=net.dtype)
        H, W = net.shape[2:]
        c = coords.float().numpy()
        for b in range(B):
            for m in range(M):
                fx, fy = int(np.floor(c[b, m, 0])), int(np.floor(c[b, m, 1]))
                for a in range(D):
                    for bb in range(D):
                        i, j = fy + a - radius, fx + bb - radius
                        if 0 <= i < H and 0 <= j < W:
                            out[b, m, :, a, bb] = net[b, :, i, j]
        return [out]
    cc.forward, cc.patchify_forward = forward, patchify_forward
    cb = types.ModuleType("cuda_ba")

    def neighbors(ii, jj):
        ix, jx = oracle.neighbors(ii.numpy(), jj.numpy())
        return [torch.from_numpy(ix), torch.from_numpy(jx)]

    def ba(poses, patches, intrinsics, target, weight, lmbda, ii, jj, kk, PPF, t0, t1, iterations, eff_impl):
        p, pt, _, _ = oracle.ba(poses[0].numpy(), patches[0].numpy(), intrinsics[0].numpy(), target[0].numpy(), weight[0].numpy(),
                                float(lmbda[0]), ii.numpy(), jj.numpy(), kk.numpy(), t0, t1, iterations, np.float32)
        poses[0].copy_(torch.from_numpy(p)); patches[0].copy_(torch.from_numpy(pt).view_as(patches[0]))
        return []
    cb.neighbors, cb.forward, cb.reproject, cb.solve_system = neighbors, ba, None, None
    return cc, cb


def main():
    cuda_to_cpu()
    for name in ("randint", "empty_like", "rand_like", "linspace", "meshgrid"):
        fn = getattr(torch, name)
        setattr(torch, name, (lambda f: lambda *a, **k: f(*a, **{**k, **({"device": "cpu"} if str(k.get("device", "")).startswith("cuda") else {})}))(fn))
    torch.cuda.synchronize = lambda *a, **k: None
    RP.load(native=cpu_native())
    from dpvo_amd.config import cfg as base_cfg, DEFAULT_YAML
    from dpvo_amd.net import VONet
    M, ht, wd = 16, 96, 128
    c = base_cfg.clone(); c.merge_from_dict(DEFAULT_YAML); c.PATCHES_PER_FRAME = M; c.BUFFER_SIZE = 64
    torch.manual_seed(7)
    ours = VONet()
    sd = {k: v.clone() for k, v in ours.state_dict().items()}
    ref_keys = set(RP.load().VONet().state_dict().keys())
    print("state-dict keys: ours - ref =", sorted(set(sd) - ref_keys), " ref - ours =", sorted(ref_keys - set(sd)))
    rcfg = RP.make_cfg(c, MIXED_PRECISION=False)
    slam = RP.make_tracker(rcfg, sd, ht, wd)
    g = torch.Generator().manual_seed(0)
    tex = torch.rand(3, ht + 64, wd + 64, generator=g)
    tex = torch.nn.functional.avg_pool2d(tex[None], 5, 1, 2)[0]
    tex = (255 * (tex - tex.min()) / (tex.max() - tex.min())).to(torch.uint8)
    intr = torch.tensor([100.0, 100.0, wd / 2, ht / 2])
    for t in range(int(sys.argv[1]) if len(sys.argv) > 1 else 11):
        torch.manual_seed(100 + t)
        RP.call(slam, float(t), tex[:, (2 * t) % 64:(2 * t) % 64 + ht, (3 * t) % 64:(3 * t) % 64 + wd], intr)
        s = RP.snapshot(slam)
        print(t, "n", s["n"], "E", s["ii"].size, "pose |t| max", float(np.abs(s["poses"][:, :3]).max()), "finite", bool(np.isfinite(s["poses"]).all()))
    # feeds
    slam2 = RP.make_tracker(rcfg, sd, ht, wd, feed_encoders=True)
    RP.feed(slam2, torch.randn(ht // 4, wd // 4, 128), torch.randn(ht // 4, wd // 4, 384))
    RP.call(slam2, 0.0, tex[:, :ht, :wd], intr)
    print("feed ok", slam2.n)


if __name__ == "__main__":
    main()

"""Trajectory-level parity: the GPU pipeline (dpvo_amd.dpvo.DPVO through the C ABI) against the CPU oracle pipeline
(oracle/dpvo_ref.py: oracle patchify / reproject / corr / update / BA chained exactly like dpvo/dpvo.py:328-473) on the same
feature maps, patch coordinates, random depths and decisions.  SURVEY.md 8(d): "ATE vs oracle trajectory"."""
import numpy as np
import pytest
import torch

from dpvo_amd import projective_ops as pops
from dpvo_amd.config import cfg as base_cfg, DEFAULT_YAML
from dpvo_amd.dpvo import DPVO
from dpvo_amd.net import VONet

pytestmark = pytest.mark.gpu


def test_trajectory_matches_oracle_pipeline(dev):
    from oracle.dpvo_ref import DPVORef
    M, ht, wd, seed = 16, 96, 128, 7
    decisions = [(True, False)] * 10 + [(True, True)] + [(True, False)] * 4 + [(True, True), (True, False)] + [(True, False)] * 3
    cfg = base_cfg.clone(); cfg.merge_from_dict(DEFAULT_YAML)
    cfg.PATCHES_PER_FRAME = M
    cfg.BUFFER_SIZE = 256
    torch.manual_seed(seed)
    net = VONet()
    slam = DPVO(cfg, net, ht=ht, wd=wd, device=dev)
    assert slam._hip_enc is not None
    sd = {k[len("update."):]: v.detach().float().cpu() for k, v in slam.network.state_dict().items() if k.startswith("update.")}
    ref = DPVORef(sd, ht, wd, M=M, BUFFER_SIZE=256, PATCH_LIFETIME=cfg.PATCH_LIFETIME, REMOVAL_WINDOW=cfg.REMOVAL_WINDOW,
                  OPTIMIZATION_WINDOW=cfg.OPTIMIZATION_WINDOW, KEYFRAME_INDEX=cfg.KEYFRAME_INDEX,
                  MOTION_DAMPING=cfg.MOTION_DAMPING, mem=slam.mem)
    captured = []
    enc = slam._hip_enc
    def hook(img, fmap_out, imap_out):                     # the encoders' outputs feed both pipelines
        enc(img, fmap_out=fmap_out, imap_out=imap_out)
        captured.append((fmap_out.float().permute(2, 0, 1).cpu().numpy(), imap_out.float().permute(2, 0, 1).cpu().numpy()))
    slam._hip_enc = hook
    state = {}
    slam.motion_probe = lambda: 1e9 if state["accept"] else 0.0
    orig = pops.motionmag_pair
    thresh = cfg.KEYFRAME_THRESH
    def fake(*a, defer=False, host_buf=None, **k):
        orig(*a, **k)
        res = (0.0, 0.0) if state["drop"] else (4 * thresh, 4 * thresh)
        return (lambda: res) if defer else res
    pops.motionmag_pair = fake
    slam.keyframe_override = lambda counter: state["drop"]       # (the one-call frame path takes the scripted decision this way)
    g = torch.Generator().manual_seed(seed)
    intr = torch.tensor([100.0, 100.0, wd / 2, ht / 2], device=dev)
    # a smooth texture translating a few pixels per frame (so that the correlation has structure)
    tex = torch.rand(3, ht + 64, wd + 64, generator=g)
    tex = torch.nn.functional.avg_pool2d(tex[None], 5, 1, 2)[0]
    tex = (255 * (tex - tex.min()) / (tex.max() - tex.min())).to(torch.uint8)
    worst_p = worst_d = worst_dmax = worst_d50 = worst_d90 = 0.0
    try:
        for t, (accept, drop) in enumerate(decisions):
            state["accept"], state["drop"] = accept, drop
            img = tex[:, (2 * t) % 64:(2 * t) % 64 + ht, (3 * t) % 64:(3 * t) % 64 + wd].contiguous().to(dev)
            x = torch.randint(1, wd // 4 - 1, (1, M), generator=g)
            y = torch.randint(1, ht // 4 - 1, (1, M), generator=g)
            coords = torch.stack([x, y], -1).float()
            depth = torch.rand(M, generator=g)
            slam(float(t), img, intr, patch_coords=coords.to(dev), depth_init=depth.to(dev))
            slam.flush()                              # (no-op unless DPVO_DEFER_KEYFRAME is set)
            fmap, imap = captured[-1]
            ref.frame(float(t), fmap, imap, coords[0].numpy(), depth.numpy(), intr.cpu().numpy(), accept, drop)
            n = slam.n
            assert n == ref.n and slam.m == ref.g.m
            assert np.array_equal(slam.pg.ii.cpu().numpy(), ref.g.ii) and np.array_equal(slam.pg.kk.cpu().numpy(), ref.g.kk)
            Pg, Pr = slam.pg.poses_[:n].cpu().numpy().astype(np.float64), ref.poses[:n].astype(np.float64)
            dg, dr = slam.pg.patches_[:n, :, 2, 1, 1].cpu().numpy().astype(np.float64), ref.patches[:n, :, 2, 1, 1].astype(np.float64)
            assert np.isfinite(Pg).all() and np.isfinite(dg).all()
            # quaternion sign is irrelevant
            sgn = np.sign((Pg[:, 3:] * Pr[:, 3:]).sum(-1, keepdims=True)); sgn[sgn == 0] = 1
            ep = np.abs(np.concatenate([Pg[:, :3] - Pr[:, :3], Pg[:, 3:] * sgn - Pr[:, 3:]], -1)).max()
            rel = np.abs(dg - dr) / np.maximum(np.abs(dr), 1e-2)
            ed = np.quantile(rel, 0.99)
            worst_p, worst_d = max(worst_p, ep), max(worst_d, ed)
            worst_d50, worst_d90 = max(worst_d50, np.quantile(rel, 0.5)), max(worst_d90, np.quantile(rel, 0.9))
            worst_dmax = max(worst_dmax, rel.max())
    finally:
        pops.motionmag_pair = orig
    print(f"trajectory parity over {len(decisions)} frames: max |pose diff| = {worst_p:.3e}, rel depth diff: "
          f"median {worst_d50:.3e}, 90th percentile {worst_d90:.3e}, 99th {worst_d:.3e}, max {worst_dmax:.3e}")
    # ATE-style figure: RMS translation difference of the final window, relative to the trajectory's extent
    ate = np.sqrt(((Pg[:, :3] - Pr[:, :3]) ** 2).sum(-1).mean())
    scale = max(np.linalg.norm(Pr[:, :3].max(0) - Pr[:, :3].min(0)), 1e-3)
    print(f"ATE(gpu vs oracle) = {ate:.3e}  (trajectory extent {scale:.3e})")
    # f16 rounding in the GEMMs / correlation perturbs the BA inputs by ~1e-3 px.  Poses are well conditioned (measured:
    # 6e-5 absolute, ATE 3e-5 on a 0.15 trajectory); depths of patches with little parallax / near-zero confidence are not
    # (Q = 1/(C + 1e-4)): measured median 3e-4, 90th percentile 2.5e-3, a tail up to 0.16 -- hence percentiles.
    assert worst_p < 1e-3 and worst_d50 < 3e-3 and worst_d90 < 2.5e-2
    assert ate < 1e-3

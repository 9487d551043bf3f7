"""Shared builders for the parity tests (seeded, CPU-generated inputs; see dpvo_amd/synthetic.py)."""
import numpy as np
import torch

from dpvo_amd import synthetic as S


def small_graph(n_frames=14, M=8):
    cfg = S.GraphCfg(M=M, REMOVAL_WINDOW=10, PATCH_LIFETIME=6)
    ii, jj, kk = S.replay_graph(n_frames, cfg)
    return ii, jj, kk, cfg


def to_cl(fmap_nchw):
    """[N,C,H,W] -> channels-last storage [N,H,W,C]"""
    return fmap_nchw.permute(0, 2, 3, 1).contiguous()


def gmap_cl(gmap):
    """[N,C,P,P] -> [N,P*P,C]"""
    N, C, P, _ = gmap.shape
    return gmap.permute(0, 2, 3, 1).reshape(N, P * P, C).contiguous()


def corr_inputs(E, n_slots=64, mem=6, H=48, W=64, seed=0, C=128, P=3):
    g = torch.Generator().manual_seed(seed)
    gmap = (torch.randn(n_slots, C, P, P, generator=g) / 4).half()
    f0 = (torch.randn(mem, C, H, W, generator=g) / 4).half()
    f1 = (torch.randn(mem, C, H // 4, W // 4, generator=g) / 4).half()
    coords = S.make_coords(E, P, H, W, seed=seed, oob_frac=0.05)
    us = torch.randint(0, n_slots, (E,), generator=g)
    vs = torch.randint(0, mem, (E,), generator=g)
    return gmap, f0, f1, coords, us, vs


def special_coords(P=3, H=48, W=64):
    """hand-made edge cases: borders, negative, far OOB, huge scale (scattered windows), non-finite"""
    off = torch.arange(P, dtype=torch.float32) - P // 2
    def patch(cx, cy, scale=1.0):
        x = cx + scale * off[None, :].expand(P, P)
        y = cy + scale * off[:, None].expand(P, P)
        return torch.stack([x, y], 0)
    cases = [
        patch(0.0, 0.0), patch(-0.5, -0.5), patch(W - 1.0, H - 1.0), patch(W + 2.3, H + 2.7), patch(1.25, H - 1.5),
        patch(-3.99, 10.0), patch(-4.0, 10.0), patch(-12.0, -12.0), patch(W + 20.0, 5.0), patch(3.0, 3.0),
        patch(20.3, 20.6, scale=3.0),      # bounding box 14x14 > 144 -> scattered path on level 0
        patch(30.1, 12.2, scale=9.0),      # scattered on both levels
        patch(1e5, 1e5), patch(-1e9, 3.0), patch(2.5e9, -7e9),
        patch(10.0, 10.0, scale=0.0),      # all nine pixels identical
        patch(15.999999, 7.000001),
    ]
    c = torch.stack(cases, 0)
    nf = c.clone()[:3]
    nf[0, 0, 1, 1] = float("nan"); nf[1, 1, 0, 2] = float("inf"); nf[2, 0, 2, 0] = -float("inf")
    return torch.cat([c, nf], 0)


def assert_close(a, b, atol, rtol, what=""):
    a = np.asarray(a, dtype=np.float64); b = np.asarray(b, dtype=np.float64)
    assert a.shape == b.shape, (what, a.shape, b.shape)
    nan_a, nan_b = np.isnan(a), np.isnan(b)
    assert (nan_a == nan_b).all(), f"{what}: NaN pattern differs ({nan_a.sum()} vs {nan_b.sum()})"
    a = np.where(nan_a, 0, a); b = np.where(nan_b, 0, b)
    err = np.abs(a - b) - (atol + rtol * np.abs(b))
    assert (err <= 0).all(), f"{what}: max abs err {np.abs(a - b).max():.3e} (atol {atol}, rtol {rtol}), worst excess {err.max():.3e}"


def golden_update_case(UpdateCls):
    """Seeded weights + inputs of tests/golden/update.npz (shared by make_golden.py, which instantiates the
    REFERENCE's Update class, and by the tests, which instantiate dpvo_amd.net.Update: same construction order,
    same RNG stream -> identical parameters; the fixture stores checksums to prove it)."""
    torch.manual_seed(1234)
    upd = UpdateCls(3).double()
    with torch.no_grad():
        for p in upd.parameters():
            if p.dim() == 1:
                p.add_(0.05 * torch.randn_like(p))
    ii, jj, kk = S.replay_graph(9, S.GraphCfg(M=4, REMOVAL_WINDOW=10, PATCH_LIFETIME=5))
    E = ii.numel()
    g = torch.Generator().manual_seed(5)
    net = torch.randn(1, E, 384, generator=g).double(); inp = torch.randn(1, E, 384, generator=g).double()
    corr = torch.randn(1, E, 882, generator=g).double()
    return upd, net, inp, corr, ii, jj, kk


def state_checksums(sd):
    return {k: float(v.double().abs().sum()) for k, v in sd.items()}

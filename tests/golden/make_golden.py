#!/usr/bin/env python
"""Generates tests/golden/*.npz by importing the REFERENCE'S OWN PYTHON (/root/reference/dpvo) in this container.

The reference cannot run as-is here (CUDA-only extensions, torch_scatter / numba / pypose / yacs absent),
so its Python layers are imported with stand-ins for the missing NATIVE dependencies only:
    cuda_corr / cuda_ba / lietorch_backends -> the CPU oracle (oracle/liboracle.so)   [what is being pinned is the
                                               reference's Python glue AROUND them: Update.forward, SoftAgg,
                                               GatedResidual, pops.transform / flow_mag / point_cloud, altcorr.patchify]
    torch_scatter                            -> scatter_softmax / scatter_sum restated from pytorch-scatter 2.1.2
                                               (torch_scatter/composite/softmax.py: max, sub, exp, sum, div)
    numba                                    -> njit = identity, so reduce_edges runs as the plain Python it is
    pypose, cv2, evo, yacs                   -> empty stubs (never called on these paths)
No reference source is copied; the script only imports it.  Run here (needs /root/reference):
    python tests/golden/make_golden.py
The .npz outputs are committed; tests/test_golden.py and the GPU tests compare against them."""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
REF = "/root/reference"

import oracle  # noqa: E402
from dpvo_amd import synthetic as S  # noqa: E402


def _t(a, like=None):
    t = torch.from_numpy(np.ascontiguousarray(a))
    return t.to(like.dtype) if like is not None else t


def install_stubs():
    # ---- torch_scatter (composite ops of pytorch-scatter 2.1.2)
    ts = types.ModuleType("torch_scatter")

    def scatter_sum(src, index, dim=-1, out=None, dim_size=None):
        assert dim == 1
        n = int(index.max()) + 1 if dim_size is None else dim_size
        o = torch.zeros(src.shape[0], n, *src.shape[2:], dtype=src.dtype)
        return o.index_add(1, index, src)

    def scatter_max(src, index, dim=-1, dim_size=None):
        n = int(index.max()) + 1 if dim_size is None else dim_size
        o = torch.full((src.shape[0], n) + tuple(src.shape[2:]), -float("inf"), dtype=src.dtype)
        idx = index.view(1, -1, *([1] * (src.dim() - 2))).expand_as(src)
        return o.scatter_reduce(1, idx, src, "amax", include_self=True), None

    def scatter_softmax(src, index, dim=-1, dim_size=None):
        assert dim == 1
        mx, _ = scatter_max(src, index, dim)
        rec = src - mx[:, index]
        ex = rec.exp()
        sm = scatter_sum(ex, index, dim)
        return ex / sm[:, index]

    ts.scatter_sum, ts.scatter_softmax, ts.scatter_max = scatter_sum, scatter_softmax, scatter_max
    sys.modules["torch_scatter"] = ts

    # ---- lietorch_backends (SE3 forward only), cuda_corr, cuda_ba on the oracle
    lb = types.ModuleType("lietorch_backends")

    def _np(x):
        return x.detach().cpu().numpy()

    def wrap1(fn):
        return lambda gid, X: _t(fn(_np(X).astype(np.float64)), X)

    def wrap2(fn):
        return lambda gid, X, Y: _t(fn(_np(X).astype(np.float64), _np(Y).astype(np.float64)), X)

    lb.expm, lb.logm, lb.inv = wrap1(oracle.se3_exp), wrap1(oracle.se3_log), wrap1(oracle.se3_inv)
    lb.mul, lb.act4 = wrap2(oracle.se3_mul), wrap2(oracle.se3_act4)

    def adjT(gid, X, a):
        """b = Adj(X)^T a, Adj = [[R, [t]x R], [0, R]] (lietorch/include/se3.h:58-67,84-86), float64 numpy"""
        x, av = _np(X).astype(np.float64), _np(a).astype(np.float64)
        t, q = x[:, :3], x[:, 3:] / np.linalg.norm(x[:, 3:], axis=1, keepdims=True)
        qx, qy, qz, qw = q.T
        R = np.stack([1 - 2 * (qy * qy + qz * qz), 2 * (qx * qy - qz * qw), 2 * (qx * qz + qy * qw),
                      2 * (qx * qy + qz * qw), 1 - 2 * (qx * qx + qz * qz), 2 * (qy * qz - qx * qw),
                      2 * (qx * qz - qy * qw), 2 * (qy * qz + qx * qw), 1 - 2 * (qx * qx + qy * qy)], -1).reshape(-1, 3, 3)
        z = np.zeros_like(qx)
        tx = np.stack([z, -t[:, 2], t[:, 1], t[:, 2], z, -t[:, 0], -t[:, 1], t[:, 0], z], -1).reshape(-1, 3, 3)
        Ad = np.zeros((x.shape[0], 6, 6))
        Ad[:, :3, :3] = R; Ad[:, :3, 3:] = tx @ R; Ad[:, 3:, 3:] = R
        return _t(np.einsum("nji,nj->ni", Ad, av), a)

    lb.adjT = adjT
    for name in ("expm_backward", "logm_backward", "inv_backward", "mul_backward", "adj", "adj_backward",
                 "adjT_backward", "act", "act_backward", "act4_backward", "Jinv", "as_matrix", "projector"):
        setattr(lb, name, None)
    sys.modules["lietorch_backends"] = lb

    cc = types.ModuleType("cuda_corr")

    def patchify_forward(net, coords, radius):
        D = 2 * radius + 2
        B, M = coords.shape[:2]
        out = torch.zeros(B, M, net.shape[1], D, D, dtype=net.dtype)
        n, c = _np(net).astype(np.float64), _np(coords).astype(np.float64)
        H, W = n.shape[2:]
        for b in range(B):
            for m in range(M):
                fx, fy = int(np.floor(c[b, m, 0])), int(np.floor(c[b, m, 1]))
                for a in range(D):
                    for bb in range(D):
                        i, j = fy + a - radius, fx + bb - radius
                        if 0 <= i < H and 0 <= j < W:
                            out[b, m, :, a, bb] = net[b, :, i, j]
        return [out]

    cc.patchify_forward = patchify_forward
    cc.forward = cc.backward = cc.patchify_backward = None
    sys.modules["cuda_corr"] = cc

    cb = types.ModuleType("cuda_ba")

    def neighbors(ii, jj):
        ix, jx = oracle.neighbors(_np(ii), _np(jj))
        return [torch.from_numpy(ix), torch.from_numpy(jx)]

    cb.neighbors = neighbors
    cb.reproject = cb.forward = cb.solve_system = None
    sys.modules["cuda_ba"] = cb

    nb = types.ModuleType("numba")
    nb.njit = lambda *a, **k: (lambda f: f)
    nb.bool_ = np.bool_
    sys.modules["numba"] = nb
    for name in ("pypose", "cv2", "evo", "yacs", "yacs.config"):
        sys.modules.setdefault(name, types.ModuleType(name))
    sys.modules["yacs.config"].CfgNode = dict
    for attr in ("SE3", "Sim3", "SO3"):                      # only referenced in type annotations at import time
        setattr(sys.modules["pypose"], attr, object)


def main():
    assert os.path.isdir(REF), "needs the reference checkout"
    install_stubs()
    sys.path.insert(0, REF)
    from dpvo import projective_ops as rpops                      # reference code
    from dpvo.lietorch import SE3 as RSE3
    from dpvo.net import Update as RUpdate
    from dpvo.altcorr.correlation import patchify as rpatchify
    from dpvo.loop_closure.optim_utils import reduce_edges as rreduce

    out = {}
    # ------------------------------------------------------------------ projective ops (f64 on the reference side)
    ii, jj, kk = S.replay_graph(14, S.GraphCfg(M=8, REMOVAL_WINDOW=10, PATCH_LIFETIME=6))
    sel = torch.arange(0, ii.numel(), 5)
    ii, jj, kk = ii[sel], jj[sel], kk[sel]
    poses, patches, intr = S.make_scene(14, M=8, ht=48, wd=64, seed=7)
    poses[:, 3:] *= 1.7                                            # un-normalised quaternions: SO3 ctor normalises
    patches[3::7, 2] *= -1                                         # behind the camera -> Z clamp
    Pd, pd, Kd = poses.double()[None], patches.double()[None], intr.double()[None]
    x1 = rpops.transform(RSE3(Pd), pd, Kd, ii, jj, kk)             # [1,E,3,3,2]
    fl, val = rpops.flow_mag(RSE3(Pd), pd, Kd, ii, jj, kk, beta=0.5)
    ix = torch.arange(14 * 8) // 8
    pc = rpops.point_cloud(RSE3(Pd), pd[:, :14 * 8], Kd, ix)
    pc = (pc[..., 1, 1, :3] / pc[..., 1, 1, 3:]).reshape(-1, 3)
    out["pops"] = dict(poses=poses.numpy(), patches=patches.numpy(), intr=intr.numpy(), ii=ii.numpy(), jj=jj.numpy(),
                       kk=kk.numpy(), coords=x1[0].permute(0, 3, 1, 2).numpy(), flow=fl[0].numpy(),
                       valid=val[0].numpy(), points=pc.numpy())

    # ------------------------------------------------------------------ Update.forward (reference module, f64, CPU)
    from tests import helpers as TH
    upd, net, inp, corr, ii2, jj2, kk2 = TH.golden_update_case(RUpdate)
    with torch.no_grad():
        n2, (d2, w2, _) = upd(net, inp, corr, None, ii2, jj2, kk2)
    cs = TH.state_checksums(upd.state_dict())
    out["update"] = dict(net_out=n2[0].float().numpy(), delta=d2[0].float().numpy(), weight=w2[0].float().numpy(),
                         ck_names=np.array(sorted(cs)), ck_vals=np.array([cs[k] for k in sorted(cs)]),
                         in_ck=np.array([float(net.abs().sum()), float(inp.abs().sum()), float(corr.abs().sum())]))

    # ------------------------------------------------------------------ altcorr.patchify bilinear glue
    g = torch.Generator().manual_seed(6)
    netp = torch.randn(1, 6, 10, 12, generator=g).double()
    cp = torch.stack([torch.rand(9, generator=g) * 14 - 1, torch.rand(9, generator=g) * 12 - 1], -1)[None].double()
    pb = rpatchify(netp, cp, 1)
    out["patchify"] = dict(net=netp[0].numpy(), coords=cp[0].numpy(), out=pb[0].numpy())

    # ------------------------------------------------------------------ reduce_edges (reference Python, numba stubbed)
    rng = np.random.default_rng(3)
    n = 300
    ri = rng.integers(0, 70, n).astype(np.int64); rj = (ri + rng.integers(0, 80, n)).astype(np.int64)
    fm = (rng.random(n) * 60).astype(np.float64)
    fm[::11] = np.inf
    es = rreduce(fm, ri, rj, 1000, 1)
    es5 = rreduce(fm, ri, rj, 5, 1)
    out["reduce_edges"] = dict(flow=fm, ii=ri, jj=rj, edges=np.asarray(es, np.int64).reshape(-1, 2),
                               edges_cap5=np.asarray(es5, np.int64).reshape(-1, 2))

    # ------------------------------------------------------------------ bundle adjustment: the reference's own PYTHON BA
    # dpvo/ba.py:86-182 (pops.transform(jacobian=True) Jacobians, torch_scatter accumulation, Schur, Cholesky, retractions)
    # is an implementation independent of ba_cuda.cu; with ep=1.0 (the CUDA damping S += I*(1e-4*S + 1), ba_cuda.cu:560),
    # the CUDA bounds (-64, -64, 2cx+64, 2cy+64) (:305-306), fixedp = t0 and residuals < 128 px it is the same update.
    from dpvo.ba import BA as rBA
    ii3, jj3, kk3 = S.replay_graph(12, S.GraphCfg(M=6, REMOVAL_WINDOW=9, PATCH_LIFETIME=5))
    poses3, patches3, intr3 = S.make_scene(12, M=6, ht=48, wd=64, seed=11)
    g = torch.Generator().manual_seed(12)
    patches3[5::9, 2] *= -1                                       # negative inverse depths (still Z > 0.2: they get updated)
    patches3[40, 0] += 300.0                                      # projects outside the bounds: masked on both sides
    patches3[47, 2] = 40.0                                        # very close point: Z < 0.2 in some target frames
    co = oracle.reproject(poses3.numpy(), patches3.numpy(), intr3.numpy(), ii3.numpy(), jj3.numpy(), kk3.numpy())
    tgt = torch.from_numpy(co[:, :, 1, 1]).double() + 1.5 * torch.randn(ii3.numel(), 2, generator=g).double()
    tgt[3::50] += 400.0                                           # residual > 250 px (> 128 px): masked on both sides
    wgt = torch.rand(ii3.numel(), 2, generator=g).double()
    t0b = 5
    cx, cy = float(intr3[0, 2]), float(intr3[0, 3])
    bounds = [-64.0, -64.0, 2 * cx + 64.0, 2 * cy + 64.0]
    Pb, pb_ = poses3.double()[None].clone(), patches3.double()[None].clone()
    Gs, ps = RSE3(Pb), pb_
    steps, tgts = [], [tgt]
    for it in range(2):
        Gs, ps = rBA(Gs, ps, intr3.double()[None], tgts[-1][None], wgt[None], 1e-4, ii3, jj3, kk3, bounds, ep=1.0, fixedp=t0b)
        steps.append((Gs.data[0].numpy().copy(), ps[0].numpy().copy()))
        # the second step starts from the first step's output with fresh targets around ITS projections, so that no
        # residual falls between the CUDA (128 px) and the Python (250 px) outlier thresholds
        co = oracle.reproject(steps[-1][0], steps[-1][1], intr3.numpy(), ii3.numpy(), jj3.numpy(), kk3.numpy())
        t2 = torch.from_numpy(co[:, :, 1, 1]).double() + 1.5 * torch.randn(ii3.numel(), 2, generator=g).double()
        t2[7::60] -= 500.0
        tgts.append(t2)
    out["ba"] = dict(poses=poses3.numpy(), patches=patches3.numpy(), intr=intr3.numpy(), target=tgt.numpy(), weight=wgt.numpy(),
                     target2=tgts[1].numpy(),
                     ii=ii3.numpy(), jj=jj3.numpy(), kk=kk3.numpy(), t0=np.int64(t0b), t1=np.int64(12),
                     poses_it1=steps[0][0], patches_it1=steps[0][1], poses_it2=steps[1][0], patches_it2=steps[1][1])

    for name, d in out.items():
        np.savez_compressed(os.path.join(HERE, f"{name}.npz"), **d)
        print(name, {k: getattr(v, "shape", None) for k, v in list(d.items())[:8]})


if __name__ == "__main__":
    main()

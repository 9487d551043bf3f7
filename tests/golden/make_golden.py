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
    for name in ("expm_backward", "logm_backward", "inv_backward", "mul_backward", "adj", "adj_backward", "adjT",
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

    for name, d in out.items():
        np.savez_compressed(os.path.join(HERE, f"{name}.npz"), **d)
        print(name, {k: getattr(v, "shape", None) for k, v in list(d.items())[:8]})


if __name__ == "__main__":
    main()

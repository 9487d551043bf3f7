"""TEST INFRASTRUCTURE ONLY -- stand-ins for the reference's third-party / un-buildable dependencies, so that the reference's OWN
Python (dpvo/dpvo.py, net.py, patchgraph.py, projective_ops.py, blocks.py, lietorch/*.py ...) can run on the MI355X around the
reference's OWN native kernels (oracle/_ref/ref_cuda_corr.so, ref_cuda_ba.so, built by oracle/build_ref.py).  Used by
oracle/ref_pipeline.py (the trajectory-level checker and the same-box reference baseline); never imported by dpvo_amd/.

What is replaced, and by what (everything here is plain torch and runs on whatever device its inputs live on):

* ``torch_scatter`` (pytorch-scatter 2.1.2, environment.yml:12; call sites blocks.py:42-43): ``scatter_max`` / ``scatter_sum`` /
  ``scatter_softmax`` restated from torch_scatter/composite/softmax.py (max, sub, exp, sum, div -- every intermediate in the dtype
  of ``src``, i.e. f16 under the tracker's autocast, as the real package does).  One deliberate difference: the group sums are
  accumulated in f32 in a FIXED order (sorted segments) and rounded once, where the real package uses f16 atomics in launch order
  -- the real thing is not run-to-run reproducible, a checker has to be.
* ``lietorch_backends`` (lietorch/src/lietorch.cpp:286-316; needs Eigen, not buildable here): the SE3 (group_id 3) forward ops
  ``expm, logm, inv, mul, adj, adjT, act, act4, as_matrix`` restated from lietorch/include/se3.h:36-142 and so3.h:31-208, incl. the
  quaternion normalisation every SO3 constructor performs (so3.h:31-37) and the EPS = 1e-6 series branches (common.h:7).
  Pinned by lietorch's own identities (run_tests.py:16-52) and against the C oracle in tests/test_oracle.py.
* ``numba`` (njit = identity: reduce_edges, optim_utils.py:23-60, then runs as the plain Python it is), ``yacs``, ``pypose``,
  ``cv2``, ``evo``, ``kornia``, ``plyfile``: empty modules (never called on the per-frame path).
"""
import sys
import types

import numpy as np
import torch

EPS = 1e-6      # lietorch/include/common.h:7


# ------------------------------------------------------------------------------------------ torch_scatter
def _bidx(index, src, dim):
    shape = [1] * src.dim()
    shape[dim] = -1
    return index.view(shape).expand_as(src)


def scatter_max(src, index, dim=-1, out=None, dim_size=None):
    dim = dim % src.dim()
    n = int(index.max()) + 1 if dim_size is None else dim_size
    shape = list(src.shape)
    shape[dim] = n
    o = torch.full(shape, -float("inf"), dtype=src.dtype, device=src.device)
    o = o.scatter_reduce(dim, _bidx(index, src, dim), src, "amax", include_self=True)
    return o, None


def scatter_sum(src, index, dim=-1, out=None, dim_size=None):
    """f32 accumulation over sorted segments (fixed order), one rounding to src.dtype"""
    dim = dim % src.dim()
    n = int(index.max()) + 1 if dim_size is None else dim_size
    x = src.movedim(dim, 0)
    order = torch.argsort(index, stable=True)
    xs = x.index_select(0, order).float().contiguous()
    counts = torch.bincount(index, minlength=n)
    try:
        red = torch.segment_reduce(xs, "sum", lengths=counts, axis=0, unsafe=True)
        red = torch.nan_to_num(red, nan=0.0) if (counts == 0).any() else red       # (empty segments; cannot happen after torch.unique)
    except (RuntimeError, NotImplementedError):
        red = torch.zeros((n,) + tuple(xs.shape[1:]), dtype=torch.float32, device=src.device).index_add_(0, index[order], xs)
    return red.to(src.dtype).movedim(0, dim)


def scatter_softmax(src, index, dim=-1, dim_size=None):
    """torch_scatter/composite/softmax.py (2.1.2): recentre by the group max, exp, normalise by the group sum"""
    dim = dim % src.dim()
    mx, _ = scatter_max(src, index, dim, dim_size=dim_size)
    rec = src - mx.index_select(dim, index)
    ex = rec.exp()
    sm = scatter_sum(ex, index, dim, dim_size=dim_size)
    return ex / sm.index_select(dim, index)


# ------------------------------------------------------------------------------------------ lietorch_backends (SE3)
def _qnorm(q):
    return q / q.norm(dim=-1, keepdim=True)                 # Eigen Quaternion::normalize (so3.h:31-37)


def _qmul(a, b):
    ax, ay, az, aw = a.unbind(-1)
    bx, by, bz, bw = b.unbind(-1)
    return torch.stack([aw * bx + ax * bw + ay * bz - az * by,
                        aw * by + ay * bw + az * bx - ax * bz,
                        aw * bz + az * bw + ax * by - ay * bx,
                        aw * bw - ax * bx - ay * by - az * bz], -1)


def _rot(q, p):
    """so3.h:55-60: p + w * uv + v x uv, uv = 2 (v x p)"""
    v, w = q[..., :3], q[..., 3:]
    uv = torch.linalg.cross(v, p[..., :3], dim=-1)
    uv = uv + uv
    return p[..., :3] + w * uv + torch.linalg.cross(v, uv, dim=-1)


def _hat(p):
    z = torch.zeros_like(p[..., 0])
    return torch.stack([z, -p[..., 2], p[..., 1], p[..., 2], z, -p[..., 0], -p[..., 1], p[..., 0], z], -1).view(p.shape[:-1] + (3, 3))


def _only_se3(gid):
    if gid != 3:
        raise NotImplementedError(f"lietorch stand-in: group_id {gid} (only SE3 = 3 is on the tracker's path)")


def se3_inv(gid, X):
    _only_se3(gid)
    t, q = X[..., :3], _qnorm(X[..., 3:])
    qi = _qnorm(q * q.new_tensor([-1, -1, -1, 1]))          # so3.inv() builds SO3(conjugate): normalised again
    return torch.cat([-_rot(qi, t), qi], -1)                # se3.h:36-38


def se3_mul(gid, X, Y):
    _only_se3(gid)
    qa, qb = _qnorm(X[..., 3:]), _qnorm(Y[..., 3:])
    return torch.cat([X[..., :3] + _rot(qa, Y[..., :3]), _qnorm(_qmul(qa, qb))], -1)    # se3.h:45-47


def se3_act(gid, X, p):
    _only_se3(gid)
    return _rot(_qnorm(X[..., 3:]), p) + X[..., :3]         # se3.h:49-51


def se3_act4(gid, X, p):
    _only_se3(gid)
    return torch.cat([_rot(_qnorm(X[..., 3:]), p) + X[..., :3] * p[..., 3:], p[..., 3:]], -1)   # se3.h:53-56


def _so3_exp(phi):
    """so3.h:152-170"""
    theta2 = (phi * phi).sum(-1, keepdim=True)
    theta = theta2.sqrt()
    theta4 = theta2 * theta2
    small = theta < EPS
    ts = torch.where(small, torch.ones_like(theta), theta)
    imag = torch.where(small, 0.5 - (1.0 / 48.0) * theta2 + (1.0 / 3840.0) * theta4, torch.sin(0.5 * ts) / ts)
    real = torch.where(small, 1 - (1.0 / 8.0) * theta2 + (1.0 / 384.0) * theta4, torch.cos(0.5 * ts))
    return _qnorm(torch.cat([imag * phi, real], -1))


def _left_jacobian(phi):
    """so3.h:172-190"""
    Phi = _hat(phi)
    Phi2 = Phi @ Phi
    theta2 = (phi * phi).sum(-1)[..., None, None]
    theta = theta2.sqrt()
    small = theta < EPS
    t2 = torch.where(small, torch.ones_like(theta2), theta2)
    t1 = t2.sqrt()
    c1 = torch.where(small, 0.5 - (1.0 / 24.0) * theta2, (1.0 - torch.cos(t1)) / t2)
    c2 = torch.where(small, (1.0 / 6.0) - (1.0 / 120.0) * theta2, (t1 - torch.sin(t1)) / (t2 * t1))
    I = torch.eye(3, dtype=phi.dtype, device=phi.device)
    return I + c1 * Phi + c2 * Phi2


def _left_jacobian_inverse(phi):
    """so3.h:192-208"""
    Phi = _hat(phi)
    Phi2 = Phi @ Phi
    theta = (phi * phi).sum(-1).sqrt()[..., None, None]
    small = theta < EPS
    t1 = torch.where(small, torch.ones_like(theta), theta)
    half = 0.5 * t1
    c2 = torch.where(small, torch.full_like(theta, 1.0 / 12.0), (1 - t1 * torch.cos(half) / (2 * torch.sin(half))) / (t1 * t1))
    I = torch.eye(3, dtype=phi.dtype, device=phi.device)
    return I - 0.5 * Phi + c2 * Phi2


def se3_exp(gid, a):
    _only_se3(gid)
    tau, phi = a[..., :3], a[..., 3:]
    return torch.cat([(_left_jacobian(phi) @ tau[..., None])[..., 0], _so3_exp(phi)], -1)      # se3.h:133-142


def _so3_log(q):
    """so3.h:114-150 (atan-based)"""
    v, w = q[..., :3], q[..., 3:]
    sq = (v * v).sum(-1, keepdim=True)
    small = sq < EPS * EPS
    n = torch.where(small, torch.ones_like(sq), sq).sqrt()
    wz = w.abs() < EPS
    ws = torch.where(wz, torch.ones_like(w), w)
    big = torch.where(wz, torch.where(w > 0, np.pi / n, -np.pi / n), 2 * torch.atan(n / ws) / n)
    wn = torch.where(small, w, torch.ones_like(w))
    sm = 2.0 / wn - (2.0 / 3.0) * sq / (wn * wn * wn)
    return torch.where(small, sm, big) * v


def se3_log(gid, X):
    _only_se3(gid)
    t, q = X[..., :3], _qnorm(X[..., 3:])
    phi = _so3_log(q)
    return torch.cat([(_left_jacobian_inverse(phi) @ t[..., None])[..., 0], phi], -1)          # se3.h:124-131


def _rotmat(q):
    qx, qy, qz, qw = q.unbind(-1)
    return torch.stack([1 - 2 * (qy * qy + qz * qz), 2 * (qx * qy - qz * qw), 2 * (qx * qz + qy * qw),
                        2 * (qx * qy + qz * qw), 1 - 2 * (qx * qx + qz * qz), 2 * (qy * qz - qx * qw),
                        2 * (qx * qz - qy * qw), 2 * (qy * qz + qx * qw), 1 - 2 * (qx * qx + qy * qy)], -1).view(q.shape[:-1] + (3, 3))


def _adjoint(X):
    """se3.h:58-67: [[R, [t]x R], [0, R]]"""
    R = _rotmat(_qnorm(X[..., 3:]))
    Ad = X.new_zeros(X.shape[:-1] + (6, 6))
    Ad[..., :3, :3] = R
    Ad[..., :3, 3:] = _hat(X[..., :3]) @ R
    Ad[..., 3:, 3:] = R
    return Ad


def se3_adj(gid, X, a):
    _only_se3(gid)
    return (_adjoint(X) @ a[..., None])[..., 0]             # se3.h:80-82


def se3_adjT(gid, X, a):
    _only_se3(gid)
    return (_adjoint(X).transpose(-1, -2) @ a[..., None])[..., 0]      # se3.h:84-86


def se3_as_matrix(gid, X):
    _only_se3(gid)
    T = torch.eye(4, dtype=X.dtype, device=X.device).repeat(X.shape[:-1] + (1, 1))
    T[..., :3, :3] = _rotmat(_qnorm(X[..., 3:]))
    T[..., :3, 3] = X[..., :3]
    return T


# ------------------------------------------------------------------------------------------ installation
def _module(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    return m


class _AttrDict(dict):
    """minimal yacs.config.CfgNode (dpvo/config.py:1-38 only assigns attributes)"""
    __getattr__ = dict.__getitem__
    __setattr__ = dict.__setitem__


def install(native=None):
    """Put the stand-ins into sys.modules (idempotent).  `native` = (ref_cuda_corr, ref_cuda_ba): the reference's own extension
    modules (oracle.ref_native()), registered under the names its Python imports them by (altcorr/correlation.py:2, fastba/ba.py:2)."""
    sys.modules["torch_scatter"] = _module("torch_scatter", scatter_sum=scatter_sum, scatter_softmax=scatter_softmax,
                                           scatter_max=scatter_max)
    lb = _module("lietorch_backends", expm=se3_exp, logm=se3_log, inv=se3_inv, mul=se3_mul, adj=se3_adj, adjT=se3_adjT,
                 act=se3_act, act4=se3_act4, as_matrix=se3_as_matrix)
    for name in ("expm_backward", "logm_backward", "inv_backward", "mul_backward", "adj_backward", "adjT_backward", "act_backward",
                 "act4_backward", "Jinv", "projector"):
        setattr(lb, name, None)                               # inference only: backward ops are never called
    sys.modules["lietorch_backends"] = lb
    nb = _module("numba", njit=lambda *a, **k: (lambda f: f), bool_=np.bool_)
    sys.modules.setdefault("numba", nb)
    for name in ("pypose", "cv2", "evo", "kornia", "plyfile", "yacs", "yacs.config"):
        sys.modules.setdefault(name, types.ModuleType(name))
    if not hasattr(sys.modules["yacs.config"], "CfgNode"):
        sys.modules["yacs.config"].CfgNode = _AttrDict
    for attr in ("SE3", "Sim3", "SO3"):                       # only referenced in type annotations at import time
        if not hasattr(sys.modules["pypose"], attr):
            setattr(sys.modules["pypose"], attr, object)
    if native is not None:
        sys.modules["cuda_corr"], sys.modules["cuda_ba"] = native

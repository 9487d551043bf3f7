"""projective_ops: the reference's `dpvo.projective_ops` surface used at inference (dpvo/projective_ops.py)
with `transform`, `flow_mag` and `point_cloud` as single fused HIP kernels (dpvo_amd/csrc/geom.hip) instead of
~12 lietorch + elementwise launches each."""
import ctypes

import torch

from . import _lib as L
from .lietorch import SE3

MIN_DEPTH = 0.2


def _prep(poses, patches, intrinsics):
    pd = poses.data if isinstance(poses, SE3) else poses
    P = patches.shape[-1]
    pd = pd.reshape(-1, 7)
    pt = patches.reshape(-1, 3, P, P)
    it = intrinsics.reshape(-1, 4)
    L.require_cuda(pd, pt, it)
    return pd.float().contiguous(), pt.float().contiguous(), it.float().contiguous(), P


def iproj(patches, intrinsics):
    """inverse projection (projective_ops.py:19-29)"""
    x, y, d = patches.unbind(dim=2)
    fx, fy, cx, cy = intrinsics[..., None, None].unbind(dim=2)
    i = torch.ones_like(d)
    xn = (x - cx) / fx
    yn = (y - cy) / fy
    return torch.stack([xn, yn, i, d], dim=-1)


def proj(X, intrinsics, depth=False):
    """projection (projective_ops.py:32-50)"""
    X, Y, Z, W = X.unbind(dim=-1)
    fx, fy, cx, cy = intrinsics[..., None, None].unbind(dim=2)
    d = 1.0 / Z.clamp(min=0.1)
    x = fx * (d * X) + cx
    y = fy * (d * Y) + cy
    if depth:
        return torch.stack([x, y, d], dim=-1)
    return torch.stack([x, y], dim=-1)


def transform_coords(poses, patches, intrinsics, ii, jj, kk):
    """Fused pops.transform + the permute of DPVO.reproject (dpvo.py:209-213): coords [1,E,2,P,P]."""
    pd, pt, it, P = _prep(poses, patches, intrinsics)
    E = ii.numel()
    coords = torch.empty(E, 2, P, P, dtype=torch.float32, device=pd.device)
    L.check(L.lib().dpvo_reproject(L.ptr(pd), L.ptr(pt), L.ptr(it), L.ptr(ii.long().contiguous()),
                                   L.ptr(jj.long().contiguous()), L.ptr(kk.long().contiguous()), L.ptr(coords),
                                   L.i64(E), L.i32(P), L.i32(1), L.stream()), "dpvo_reproject")
    return coords.view(1, E, 2, P, P)


def transform(poses, patches, intrinsics, ii, jj, kk, depth=False, valid=False, jacobian=False, tonly=False):
    """projective transform (projective_ops.py:53-112), inference subset: returns x1 [1,E,P,P,2]."""
    if depth or valid or jacobian or tonly:
        raise NotImplementedError("transform(depth/valid/jacobian/tonly) is training / flow_mag-internal only; "
                                  "use flow_mag() or the fused kernels")
    return transform_coords(poses, patches, intrinsics, ii, jj, kk).permute(0, 1, 3, 4, 2)


def point_cloud(poses, patches, intrinsics, ix, out=None):
    """Centre-pixel 3-D points X/W, i.e. exactly what dpvo.py:358-360 keeps of pops.point_cloud: [m,3].
    `out`: optional contiguous f32 [>=m,3] buffer to write into (its first m rows are returned)."""
    pd, pt, it, P = _prep(poses, patches, intrinsics)
    m = ix.numel()
    assert pt.shape[0] >= m
    if out is not None:
        assert out.is_contiguous() and out.dtype == torch.float32 and out.shape[0] >= m and out.shape[1] == 3
        points = out[:m]
    else:
        points = torch.empty(m, 3, dtype=torch.float32, device=pd.device)
    L.check(L.lib().dpvo_point_cloud(L.ptr(pd), L.ptr(pt), L.ptr(it), L.ptr(ix.long().contiguous()), L.ptr(points),
                                     L.i64(m), L.i32(P), L.stream()), "dpvo_point_cloud")
    return points


def flow_mag(poses, patches, intrinsics, ii, jj, kk, beta=0.3):
    """pops.flow_mag (projective_ops.py:120-130) reduced over the patch: (mean flow [E], #valid pixels [E])."""
    pd, pt, it, P = _prep(poses, patches, intrinsics)
    E = ii.numel()
    flow = torch.empty(E, dtype=torch.float32, device=pd.device)
    val = torch.empty(E, dtype=torch.float32, device=pd.device)
    L.check(L.lib().dpvo_flow_mag(L.ptr(pd), L.ptr(pt), L.ptr(it), L.ptr(ii.long().contiguous()),
                                  L.ptr(jj.long().contiguous()), L.ptr(kk.long().contiguous()), L.f32(beta), L.ptr(flow),
                                  L.ptr(val), L.i64(E), L.i32(P), L.stream()), "dpvo_flow_mag")
    return flow, val


def motionmag_pair(poses, patches, intrinsics, ii, jj, kk, i, j, beta=0.5, plan=None, defer=False, host_buf=None):
    """DPVO.motionmag(i,j) + DPVO.motionmag(j,i) (dpvo.py:257-264,269) as ONE kernel + ONE host read-back.
    Returns (mean flow i->j, mean flow j->i) as Python floats (NaN when a direction has no edge); with defer=True
    returns a callable that performs the read-back, so that the caller can do host work while the kernel runs."""
    pd, pt, it, P = _prep(poses, patches, intrinsics)
    E = ii.numel()
    with_plan = plan is not None and plan.E == E and E > 0
    status = with_plan and (host_buf is None or host_buf.numel() >= 8)
    out = torch.empty(8 if status else 4, dtype=torch.float32, device=pd.device)
    if status:        # the plan's counters + window-violation flag come back with the same read-back (out[4:8])
        L.check(L.lib().dpvo_motionmag_status(L.ptr(pd), L.ptr(pt), L.ptr(it), L.ptr(ii), L.ptr(jj), L.ptr(kk), L.ptr(plan.buf),
                                              L.i64(E), L.i32(P), L.i64(i), L.i64(j), L.f32(beta), L.ptr(out),
                                              ctypes.c_void_p(out.data_ptr() + 16), L.stream()), "dpvo_motionmag_status")
    else:
        L.check(L.lib().dpvo_motionmag(L.ptr(pd), L.ptr(pt), L.ptr(it), L.ptr(ii), L.ptr(jj), L.ptr(kk),
                                       L.ptr(plan.buf if with_plan else None), L.i64(E), L.i32(P),
                                       L.i64(i), L.i64(j), L.f32(beta), L.ptr(out), L.stream()), "dpvo_motionmag")
    ev = None
    if host_buf is not None:            # pinned host buffer: asynchronous copy + event instead of a stream-wide sync
        host_buf[:out.numel()].copy_(out, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()

    def finish():
        if ev is not None:
            ev.synchronize()
            vals = host_buf[:out.numel()].tolist()
        else:
            vals = out.tolist()
        s0, n0, s1, n1 = vals[:4]
        finish.plan_status = tuple(int(v) for v in vals[4:8]) if status else None     # (n_patches, n_pairs, 0, violation flag)
        nan = float("nan")
        return (s0 / n0 if n0 > 0 else nan), (s1 / n1 if n1 > 0 else nan)
    return finish if defer else finish()

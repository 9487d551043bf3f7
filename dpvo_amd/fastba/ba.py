"""fastba: drop-in for the reference's `dpvo.fastba` (dpvo/fastba/ba.py:4-8) on the HIP kernels of
dpvo_amd/csrc/ba.hip, graph.hip and geom.hip."""
import ctypes

import torch

from .. import _lib as L
from .. import workspace
from ..graph import GraphPlan


def neighbors(kk, jj):
    """cuda_ba.neighbors (ba.cpp:59-97): (ix, jx) int64 device tensors, computed on the device."""
    L.require_cuda(kk, jj)
    E = kk.numel()
    kk = kk.long().contiguous(); jj = jj.long().contiguous()
    ix = torch.empty(E, dtype=torch.long, device=kk.device)
    jx = torch.empty(E, dtype=torch.long, device=kk.device)
    nbytes = L.lib().dpvo_neighbors_workspace_bytes(L.i64(E))
    ws = workspace.get(nbytes, kk.device, "plan")
    L.check(L.lib().dpvo_neighbors(L.ptr(kk), L.ptr(jj), L.ptr(ix), L.ptr(jx), L.i64(E), L.ptr(ws),
                                   ctypes.c_size_t(ws.numel()), L.stream()), "dpvo_neighbors")
    return ix, jx


def reproject(poses, patches, intrinsics, ii, jj, kk, clamp_z=False):
    """cuda_ba.reproject (ba.cpp:48-56, ba_cuda.cu:379-429,585-615): coords [1,N,2,P,P].
    clamp_z=True gives pops.transform's semantics instead (see projective_ops.transform)."""
    L.require_cuda(poses, patches, intrinsics, ii, jj, kk)
    P = patches.shape[-1]
    E = ii.numel()
    poses = poses.reshape(-1, 7).float().contiguous()
    patches = patches.reshape(-1, 3, P, P).float().contiguous()
    intrinsics = intrinsics.reshape(-1, 4).float().contiguous()
    coords = torch.empty(E, 2, P, P, dtype=torch.float32, device=poses.device)
    L.check(L.lib().dpvo_reproject(L.ptr(poses), L.ptr(patches), L.ptr(intrinsics), L.ptr(ii.long().contiguous()),
                                   L.ptr(jj.long().contiguous()), L.ptr(kk.long().contiguous()), L.ptr(coords),
                                   L.i64(E), L.i32(P), L.i32(1 if clamp_z else 0), L.stream()), "dpvo_reproject")
    return coords.view(1, E, 2, P, P)


def BA(poses, patches, intrinsics, target, weight, lmbda, ii, jj, kk, t0, t1, M, iterations, eff_impl=False, plan=None,
       info=None, f0=None, n_frames=None):
    """cuda_ba.forward (ba.py:7-8, ba_cuda.cu:433-582): updates `poses` and `patches` storage IN PLACE, returns [].

    `plan` (a GraphPlan of the same ii,jj,kk) may be passed to reuse the per-frame index structures."""
    poses = getattr(poses, "data", poses)
    L.require_cuda(poses, patches, intrinsics, target, weight, ii, jj, kk)
    for t in (poses, patches, intrinsics):
        if t.dtype != torch.float32 or not t.is_contiguous():
            raise L.DPVOHipError("BA updates poses/patches in place: they must be contiguous float32 tensors")
    P = patches.shape[-1]
    E = ii.numel()
    N = t1 - t0
    if eff_impl or 6 * N > 120:
        from .global_ba import global_BA
        # (f0 / n_frames: a frame range that covers every source frame, from the caller's bookkeeping: no read-back of min / max kk)
        return global_BA(poses, patches, intrinsics, target, weight, lmbda, ii, jj, kk, t0, t1, M, iterations, plan=plan, f0=f0,
                         n_frames=n_frames)
    ii = ii.long().contiguous(); jj = jj.long().contiguous(); kk = kk.long().contiguous()
    target = target.reshape(-1, 2).float().contiguous()
    weight = weight.reshape(-1, 2).float().contiguous()
    if plan is None:
        plan = GraphPlan(ii, jj, kk)
    lm = float(lmbda) if not torch.is_tensor(lmbda) else float(lmbda.reshape(-1)[0].item())
    nbytes = L.lib().dpvo_ba_workspace_bytes(L.i64(E), L.i32(N))
    ws = workspace.get(nbytes, poses.device, "ba")
    L.check(L.lib().dpvo_ba(L.ptr(poses), L.ptr(patches), L.ptr(intrinsics), L.ptr(target), L.ptr(weight), L.f32(lm),
                            L.ptr(ii), L.ptr(jj), L.ptr(kk), L.ptr(plan.buf), L.i64(plan.n_patches_host),
                            L.i64(plan.n_pairs_host), L.i64(E), L.i32(P), L.i32(t0), L.i32(t1),
                            L.i32(iterations), L.ptr(info), L.ptr(ws), ctypes.c_size_t(ws.numel()), L.stream()),
            "dpvo_ba")
    return []


def solve_system(J_Ginv_i, J_Ginv_j, ii, jj, res, ep, lm, freen):
    """cuda_ba.solve_system (ba.cpp:120-180, called by loop_closure/optim_utils.py:229): one Levenberg-Marquardt step of the Sim(3)
    pose-graph optimisation -> [delta [n, 7]] (a one-element list, like the pybind entry returns).  J_Ginv_i / J_Ginv_j [r,7,7], ii / jj
    [r] long, res [r,7]; n = max(ii, jj) + 1 (read back, as the reference does at :131).  Assembled and solved in f64 on the device
    (csrc/pgo.hip).  Raises where the reference calls exit(1) (an edge with ii == jj) and when the damped system is not positive
    definite (the reference's Eigen solve would return garbage silently)."""
    L.require_cuda(J_Ginv_i, J_Ginv_j, ii, jj, res)
    Ji = J_Ginv_i.reshape(-1, 7, 7).float().contiguous()
    Jj = J_Ginv_j.reshape(-1, 7, 7).float().contiguous()
    rs = res.reshape(-1, 7).float().contiguous()
    ii = ii.long().contiguous(); jj = jj.long().contiguous()
    r = ii.numel()
    if not (Ji.shape[0] == Jj.shape[0] == rs.shape[0] == jj.numel() == r) or r == 0:
        raise L.DPVOHipError("solve_system: J_Ginv_i, J_Ginv_j [r,7,7], ii, jj [r], res [r,7] with r > 0")
    n = int(torch.maximum(ii.max(), jj.max()).item()) + 1
    freen = int(freen)
    delta = torch.empty(n, 7, dtype=torch.float32, device=rs.device)
    info = torch.zeros(1, dtype=torch.int32, device=rs.device)
    nbytes = L.lib().dpvo_solve_system_workspace_bytes(L.i64(n), L.i64(freen))
    ws = workspace.get(nbytes, rs.device, "pgo")
    L.check(L.lib().dpvo_solve_system(L.ptr(Ji), L.ptr(Jj), L.ptr(ii), L.ptr(jj), L.ptr(rs), L.i64(r), L.i64(n), L.f32(float(ep)),
                                      L.f32(float(lm)), L.i64(freen), L.ptr(delta), L.ptr(info), L.ptr(ws), ctypes.c_size_t(ws.numel()),
                                      L.stream()), "dpvo_solve_system")
    code = int(info.item())
    if code == 1:
        raise L.DPVOHipError("solve_system: an edge connects a node with itself (the reference exits the process here, ba.cpp:139-140)")
    if code == 2:
        raise L.DPVOHipError("solve_system: the damped normal equations are not positive definite")
    return [delta]

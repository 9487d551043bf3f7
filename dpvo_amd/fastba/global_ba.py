"""Global bundle adjustment (reference: eff_impl=True, dpvo/fastba/block_e.cu:43-300, ba_cuda.cu:538-550; caller
DPVO.__run_global_BA, dpvo/dpvo.py:312-326).

Routed here when `eff_impl=True` or when more than 20 poses are free (beyond the in-LDS dense Schur path).
Linearisation, block-sparse Schur complement and the retractions are HIP kernels (dpvo_amd/csrc/ba_global.hip); the
6N x 6N damped system is solved by the blocked device Cholesky of dpvo_amd/csrc/chol.hip (`dpvo_gba_solve`: damping,
factorisation and both substitutions; the reference calls cuSOLVER through ATen there, ba_cuda.cu:546-548)."""
import ctypes

import torch

from .. import _lib as L
from .. import workspace
from ..graph import GraphPlan


_PROFILE = None          # tools/gba_bench.py sets this to a list: (phase name, start event, end event) per phase


def _mark(name, ev):
    if _PROFILE is not None:
        e = torch.cuda.Event(enable_timing=True)
        e.record()
        _PROFILE.append((name, ev, e))
        ev = torch.cuda.Event(enable_timing=True)
        ev.record()
    return ev


def global_BA(poses, patches, intrinsics, target, weight, lmbda, ii, jj, kk, t0, t1, M, iterations, plan=None, f0=None,
              n_frames=None):
    """f0 / n_frames: first source frame that owns a patch with an edge and the number of frames up to the last one; the
    tracker knows them from its own bookkeeping (no read-back); computed here (one host sync) when not given."""
    P = patches.shape[-1]
    E = ii.numel()
    N = t1 - t0
    if E == 0 or N <= 0:
        raise L.DPVOHipError("global BA needs edges and at least one free pose")
    ii = ii.long().contiguous(); jj = jj.long().contiguous(); kk = kk.long().contiguous()
    target = target.reshape(-1, 2).float().contiguous()
    weight = weight.reshape(-1, 2).float().contiguous()
    if plan is None:
        plan = GraphPlan(ii, jj, kk)
    if f0 is None or n_frames is None:
        kmin, kmax = (int(v) for v in torch.stack([kk.min(), kk.max()]).tolist())     # one read-back
        f0 = kmin // M
        n_frames = kmax // M - f0 + 1
    lm = float(lmbda) if not torch.is_tensor(lmbda) else float(lmbda.reshape(-1)[0].item())
    nbytes = L.lib().dpvo_gba_workspace_bytes(L.i64(E), L.i64(plan.n_pairs_host), L.i64(n_frames), L.i32(M), L.i64(N))
    ws = workspace.get(nbytes, poses.device, "gba")
    n6 = 6 * N
    dev = poses.device
    # the damped system and its right-hand side live in a per-stream scratch buffer (no allocation per iteration)
    sy = workspace.get((n6 * n6 + 2 * n6) * 4, dev, "gba_sys").view(torch.float32)
    S = sy[:n6 * n6].view(n6, n6)
    y = sy[n6 * n6:n6 * n6 + n6]
    dX = sy[n6 * n6 + n6:n6 * n6 + 2 * n6]
    cws_bytes = L.lib().dpvo_gba_solve_workspace_bytes(L.i32(n6))
    cws = workspace.get(cws_bytes, dev, "gba_chol")
    ev = None
    if _PROFILE is not None:
        ev = torch.cuda.Event(enable_timing=True)
        ev.record()
    for it in range(iterations):
        sy[:n6 * n6 + n6].zero_()
        # (iterations after the first reuse the index structures the first one left in `ws`: dpvo_gba_relinearize)
        L.check((L.lib().dpvo_gba_linearize if it == 0 else L.lib().dpvo_gba_relinearize)(
            L.ptr(poses), L.ptr(patches), L.ptr(intrinsics), L.ptr(target), L.ptr(weight), L.f32(lm), L.ptr(ii), L.ptr(jj),
            L.ptr(kk), L.ptr(plan.buf), L.i64(plan.n_patches_host), L.i64(plan.n_pairs_host), L.i64(E), L.i32(P), L.i32(M),
            L.i32(f0), L.i32(n_frames), L.i32(t0), L.i32(t1), L.ptr(S), L.ptr(y), L.ptr(ws), ctypes.c_size_t(ws.numel()),
            L.stream()), "dpvo_gba_linearize")
        ev = _mark("linearise + Schur", ev)
        # S += I * (1e-4 * S + 1.0); U = cholesky(S); dX = cholesky_solve(y, U)   (ba_cuda.cu:546-548)
        L.check(L.lib().dpvo_gba_solve(L.ptr(S), L.ptr(y), L.i32(n6), L.ptr(dX), L.ptr(cws), ctypes.c_size_t(cws_bytes),
                                       L.stream()), "dpvo_gba_solve")
        ev = _mark("damping + Cholesky + substitutions (chol.hip)", ev)
        L.check(L.lib().dpvo_gba_retract(
            L.ptr(poses), L.ptr(patches), L.ptr(plan.buf), L.i64(plan.n_patches_host), L.i64(plan.n_pairs_host), L.i64(E),
            L.i32(P), L.i32(M), L.i32(f0), L.i32(n_frames), L.i32(t0), L.i32(t1), L.ptr(dX), L.ptr(ws),
            ctypes.c_size_t(ws.numel()), L.stream()), "dpvo_gba_retract")
        ev = _mark("back-substitution + retraction", ev)
    return []

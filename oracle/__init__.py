"""CPU oracle for the DPVO hot path -- TEST INFRASTRUCTURE ONLY.

ctypes/numpy front-end of ``oracle/liboracle.so`` (built from ``dpvo_oracle.c`` by ``make -C oracle``)
plus the torch-CPU restatement of the update operator (``oracle.update_ref``).

Only ``tests/``, ``bench.py``'s ``cpu_baseline`` leg and ``__graft_entry__.smoke()`` may import this
package, and only as the checker.  Nothing under ``dpvo_amd/`` imports it.

Parity status: the reference ships no golden vectors for altcorr / fastba / Update.  What pins this
oracle: (1) goldens made by importing the reference's Python (tests/golden/), (2) since round 3 the reference's OWN
native kernels, compiled for gfx950 by ``oracle/build_ref.py`` into ``oracle/_ref`` (``ref_native()`` below) and run on
the GPU box beside the HIP library and this restatement (tests/test_gpu_ref.py); see ``dpvo_oracle.c`` header.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "liboracle.so")


def build(force=False):
    src = [os.path.join(_HERE, f) for f in ("dpvo_oracle.c", "oracle_body.inc", "Makefile")]
    if force or not os.path.exists(_SO) or any(os.path.getmtime(s) > os.path.getmtime(_SO) for s in src):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _SO


_lib = None
_ref_mods = None


def ref_native():
    """(ref_cuda_corr, ref_cuda_ba): the reference's native extensions (cuda_corr: correlation.cpp:57-63, cuda_ba:
    ba.cpp:183-189 minus solve_system) built from /root/reference by oracle/build_ref.py, or None when oracle/_ref
    is absent.  They are torch extension modules and need a GPU to run."""
    global _ref_mods
    if _ref_mods is None:
        import importlib.machinery
        import importlib.util
        import torch  # noqa: F401  (the extensions link against libtorch)
        mods = []
        for name in ("ref_cuda_corr", "ref_cuda_ba"):
            path = os.path.join(_HERE, "_ref", name + ".so")
            if not os.path.exists(path):
                return None
            loader = importlib.machinery.ExtensionFileLoader(name, path)
            spec = importlib.util.spec_from_loader(name, loader)
            mod = importlib.util.module_from_spec(spec)
            loader.exec_module(mod)
            mods.append(mod)
        _ref_mods = tuple(mods)
    return _ref_mods


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        _lib = ctypes.CDLL(_SO)
        _lib.orc_unique.restype = ctypes.c_int64
        _lib.orc_reduce_edges.restype = ctypes.c_int64
        for s in ("_f32", "_f64"):
            getattr(_lib, "orc_ba" + s).restype = ctypes.c_int
            getattr(_lib, "orc_ba_full_solve" + s).restype = ctypes.c_int
    return _lib


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def _real(dtype):
    dtype = np.dtype(dtype)
    if dtype == np.float32:
        return "_f32", np.float32, ctypes.c_float
    if dtype == np.float64:
        return "_f64", np.float64, ctypes.c_double
    raise TypeError(dtype)


def _c(a, dt):
    return np.ascontiguousarray(a, dtype=dt)


def _i64(a):
    return np.ascontiguousarray(a, dtype=np.int64)


# ------------------------------------------------------------------ SE3
def se3_inv(X, dtype=np.float64):
    s, dt, _ = _real(dtype)
    X = _c(X, dt); Y = np.empty_like(X)
    getattr(lib(), "orc_se3_inv" + s)(_p(X), _p(Y), ctypes.c_int64(X.size // 7))
    return Y


def se3_mul(X, Y, dtype=np.float64):
    s, dt, _ = _real(dtype)
    X = _c(X, dt); Y = _c(Y, dt); Z = np.empty_like(X)
    getattr(lib(), "orc_se3_mul" + s)(_p(X), _p(Y), _p(Z), ctypes.c_int64(X.size // 7))
    return Z


def se3_act4(X, P, dtype=np.float64):
    s, dt, _ = _real(dtype)
    X = _c(X, dt); P = _c(P, dt); Q = np.empty_like(P)
    getattr(lib(), "orc_se3_act4" + s)(_p(X), _p(P), _p(Q), ctypes.c_int64(X.size // 7))
    return Q


def se3_exp(A, dtype=np.float64):
    s, dt, _ = _real(dtype)
    A = _c(A, dt); X = np.empty(A.shape[:-1] + (7,), dt)
    getattr(lib(), "orc_se3_exp" + s)(_p(A), _p(X), ctypes.c_int64(A.size // 6))
    return X


def se3_log(X, dtype=np.float64):
    s, dt, _ = _real(dtype)
    X = _c(X, dt); A = np.empty(X.shape[:-1] + (6,), dt)
    getattr(lib(), "orc_se3_log" + s)(_p(X), _p(A), ctypes.c_int64(X.size // 7))
    return A


# ------------------------------------------------------------------ projective ops
def reproject(poses, patches, intrinsics, ii, jj, kk, dtype=np.float64):
    """pops.transform as used by DPVO.reproject -> coords [E,2,P,P]."""
    s, dt, _ = _real(dtype)
    poses = _c(poses, dt).reshape(-1, 7); intr = _c(intrinsics, dt).reshape(-1, 4)
    P = patches.shape[-1]
    patches = _c(patches, dt).reshape(-1, 3, P, P)
    ii, jj, kk = _i64(ii), _i64(jj), _i64(kk)
    E = ii.size
    out = np.empty((E, 2, P, P), dt)
    getattr(lib(), "orc_reproject" + s)(_p(poses), _p(patches), _p(intr), _p(ii), _p(jj), _p(kk),
                                        ctypes.c_int64(E), ctypes.c_int(P), _p(out))
    return out


def point_cloud(poses, patches, intrinsics, ix, dtype=np.float64):
    s, dt, _ = _real(dtype)
    poses = _c(poses, dt).reshape(-1, 7); intr = _c(intrinsics, dt).reshape(-1, 4)
    P = patches.shape[-1]
    patches = _c(patches, dt).reshape(-1, 3, P, P)
    ix = _i64(ix)
    out = np.empty((ix.size, 3), dt)
    getattr(lib(), "orc_point_cloud" + s)(_p(poses), _p(patches), _p(intr), _p(ix), ctypes.c_int64(ix.size),
                                          ctypes.c_int(P), _p(out))
    return out


def flow_mag(poses, patches, intrinsics, ii, jj, kk, beta=0.3, dtype=np.float64):
    s, dt, cr = _real(dtype)
    poses = _c(poses, dt).reshape(-1, 7); intr = _c(intrinsics, dt).reshape(-1, 4)
    P = patches.shape[-1]
    patches = _c(patches, dt).reshape(-1, 3, P, P)
    ii, jj, kk = _i64(ii), _i64(jj), _i64(kk)
    E = ii.size
    flow = np.empty((E, P, P), dt); val = np.empty((E, P, P), dt)
    getattr(lib(), "orc_flow_mag" + s)(_p(poses), _p(patches), _p(intr), _p(ii), _p(jj), _p(kk),
                                       ctypes.c_int64(E), ctypes.c_int(P), cr(beta), _p(flow), _p(val))
    return flow, val > 0.5


# ------------------------------------------------------------------ altcorr
def corr_forward(fmap1, fmap2, coords, us, vs, radius, dtype=np.float64):
    """cuda_corr.forward for batch 1.  fmap1 [N1,C,P,P], fmap2 [N2,C,H2,W2], coords [E,2,P,P].
    Returns the reference's return value layout [E, D-1 (x), D-1 (y), P, P] (after its permute)."""
    s, dt, _ = _real(dtype)
    f1 = _c(fmap1, dt); f2 = _c(fmap2, dt); co = _c(coords, dt)
    us, vs = _i64(us), _i64(vs)
    E = us.size
    C, P = f1.shape[1], f1.shape[2]
    H2, W2 = f2.shape[2], f2.shape[3]
    D = 2 * radius + 2
    out = np.empty((E, D - 1, D - 1, P, P), dt)
    getattr(lib(), "orc_corr_forward" + s)(_p(f1), _p(f2), _p(co), _p(us), _p(vs), ctypes.c_int64(E),
                                           ctypes.c_int(C), ctypes.c_int(P), ctypes.c_int(H2), ctypes.c_int(W2),
                                           ctypes.c_int(radius), _p(out))
    return out.transpose(0, 2, 1, 3, 4)


def corr_forward_h16(fmap1, fmap2, coords, us, vs, radius):
    """cuda_corr.forward on HALF features in the reference's own arithmetic (every product, partial sum and blend step
    rounded to f16: correlation_kernel.cu:121-131,221-230).  Inputs are rounded to f16 first; returns float32 values that
    are f16-representable, reference layout [E, D-1 (x), D-1 (y), P, P]."""
    f1 = _c(np.asarray(fmap1, np.float16), np.float32); f2 = _c(np.asarray(fmap2, np.float16), np.float32)
    co = _c(coords, np.float32)
    us, vs = _i64(us), _i64(vs)
    E = us.size
    C, P = f1.shape[1], f1.shape[2]
    H2, W2 = f2.shape[2], f2.shape[3]
    D = 2 * radius + 2
    out = np.empty((E, D - 1, D - 1, P, P), np.float32)
    lib().orc_corr_forward_h16(_p(f1), _p(f2), _p(co), _p(us), _p(vs), ctypes.c_int64(E), ctypes.c_int(C), ctypes.c_int(P),
                               ctypes.c_int(H2), ctypes.c_int(W2), ctypes.c_int(radius), _p(out))
    return out.transpose(0, 2, 1, 3, 4)


def corr_pyramid(gmap, pyramid, coords, ii1, jj1, radius=3, levels=(1, 4), dtype=np.float64, emulate_f16=False):
    """DPVO.corr (dpvo.py:200-207): two levels stacked on the last axis and flattened -> [E, 2*49*P*P].
    emulate_f16: the reference's half arithmetic (corr_forward_h16); the level-1 coordinates are coords / 4 in f32 as at
    dpvo.py:206."""
    outs = []
    for f2, lvl in zip(pyramid, levels):
        if emulate_f16:
            outs.append(corr_forward_h16(gmap, f2, np.asarray(coords, np.float32) / np.float32(lvl), ii1, jj1, radius))
        else:
            outs.append(corr_forward(gmap, f2, np.asarray(coords, dtype) / lvl, ii1, jj1, radius, dtype))
    return np.stack(outs, -1).reshape(len(ii1), -1)


def patchify(net, coords, radius, dtype=np.float64):
    """altcorr.patchify(net[C,H,W], coords[M,2], radius, mode='bilinear') -> [M,C,d,d]."""
    s, dt, _ = _real(dtype)
    net = _c(net, dt); co = _c(coords, dt)
    C, H, W = net.shape
    M = co.shape[0]
    d = 2 * radius + 1
    out = np.empty((M, C, d, d), dt)
    getattr(lib(), "orc_patchify" + s)(_p(net), _p(co), ctypes.c_int64(M), ctypes.c_int(C), ctypes.c_int(H),
                                       ctypes.c_int(W), ctypes.c_int(radius), _p(out))
    return out


# ------------------------------------------------------------------ integer bookkeeping
def unique(x):
    x = _i64(x)
    u = np.empty_like(x); inv = np.empty_like(x)
    m = lib().orc_unique(_p(x), ctypes.c_int64(x.size), _p(u), _p(inv))
    return u[:m].copy(), inv


def neighbors(kk, jj):
    kk, jj = _i64(kk), _i64(jj)
    ix = np.empty_like(kk); jx = np.empty_like(kk)
    lib().orc_neighbors(_p(kk), _p(jj), ctypes.c_int64(kk.size), _p(ix), _p(jx))
    return ix, jx


def reduce_edges(flow_mag_, ii, jj, max_num_edges, nms):
    fm = _c(flow_mag_, np.float64); ii, jj = _i64(ii), _i64(jj)
    order = _i64(np.argsort(fm))
    out = np.empty((max_num_edges + 1, 2), np.int64)
    n = lib().orc_reduce_edges(_p(fm), _p(ii), _p(jj), _p(order), ctypes.c_int64(fm.size),
                               ctypes.c_int64(max_num_edges), ctypes.c_int64(nms), _p(out))
    return out[:n].copy()


# ------------------------------------------------------------------ fastba
def ba(poses, patches, intrinsics, target, weight, lmbda, ii, jj, kk, t0, t1, iterations=2, dtype=np.float64):
    """cuda_ba.forward: returns UPDATED copies (poses, patches, info, r_totals)."""
    s, dt, cr = _real(dtype)
    poses = _c(poses, dt).reshape(-1, 7).copy(); intr = _c(intrinsics, dt).reshape(-1, 4)
    P = patches.shape[-1]
    patches = _c(patches, dt).reshape(-1, 3, P, P).copy()
    target = _c(target, dt).reshape(-1, 2); weight = _c(weight, dt).reshape(-1, 2)
    ii, jj, kk = _i64(ii), _i64(jj), _i64(kk)
    kx, ku = unique(kk)
    rt = np.zeros(iterations, np.float64)
    info = getattr(lib(), "orc_ba" + s)(_p(poses), _p(patches), _p(intr), _p(target), _p(weight), cr(lmbda),
                                        _p(ii), _p(jj), _p(kk), _p(kx), _p(ku), ctypes.c_int64(ii.size),
                                        ctypes.c_int64(kx.size), ctypes.c_int(P), ctypes.c_int(t0),
                                        ctypes.c_int(t1), ctypes.c_int(iterations), _p(rt))
    return poses, patches, info, rt


def ba_full_solve(poses, patches, intrinsics, target, weight, lmbda, ii, jj, kk, t0, t1, dtype=np.float64):
    s, dt, cr = _real(dtype)
    poses = _c(poses, dt).reshape(-1, 7); intr = _c(intrinsics, dt).reshape(-1, 4)
    P = patches.shape[-1]
    patches = _c(patches, dt).reshape(-1, 3, P, P)
    target = _c(target, dt).reshape(-1, 2); weight = _c(weight, dt).reshape(-1, 2)
    ii, jj, kk = _i64(ii), _i64(jj), _i64(kk)
    kx, ku = unique(kk)
    dX = np.empty(6 * (t1 - t0), dt); dZ = np.empty(kx.size, dt)
    info = getattr(lib(), "orc_ba_full_solve" + s)(_p(poses), _p(patches), _p(intr), _p(target), _p(weight),
                                                   cr(lmbda), _p(ii), _p(jj), _p(kk), _p(ku),
                                                   ctypes.c_int64(ii.size), ctypes.c_int64(kx.size),
                                                   ctypes.c_int(P), ctypes.c_int(t0), ctypes.c_int(t1), _p(dX), _p(dZ))
    return dX, dZ, kx, info

"""The three stand-in modules a maintainer of the REFERENCE would add to run its own, unmodified Python on `libdpvo_hip.so`.

The reference reaches native code at exactly three import sites -- `import cuda_corr` (dpvo/altcorr/correlation.py:2),
`import cuda_ba` (dpvo/fastba/ba.py:2), `import lietorch_backends` (dpvo/lietorch/group_ops.py:1), the three pybind11 extensions of
setup.py:13-36.  The classes below carry the same names and call signatures as those modules and forward every call to the C ABI of
include/dpvo_hip.h through `ctypes`; nothing else of this package is imported (no `dpvo_amd._lib`, no `dpvo_amd.dpvo`): this file IS the
integration INTEGRATION.md §1-3 describes, and tests/test_gpu_integration_stubs.py runs the reference's own `DPVO` class on top of it.

    import dpvo_amd.integration_stubs as S
    sys.modules["cuda_corr"], sys.modules["cuda_ba"], sys.modules["lietorch_backends"] = S.cuda_corr, S.cuda_ba, S.lietorch_backends
    import dpvo.dpvo            # the reference, unchanged

Scope: the inference path (forward entries; the reference's backward kernels are training-only), SE3 (group_id 3) for lietorch.
"""
import ctypes
import os

import torch

_L = ctypes.CDLL(os.environ.get("DPVO_HIP_LIB") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "libdpvo_hip.so"))
for _n in ("dpvo_neighbors_workspace_bytes", "dpvo_plan_workspace_bytes", "dpvo_ba_workspace_bytes", "dpvo_solve_system_workspace_bytes"):
    getattr(_L, _n).restype = ctypes.c_size_t            # (every other entry returns int: 0 ok, < 0 DPVO_E_*, > 0 hipError_t)
_p = lambda t: ctypes.c_void_p(t.data_ptr())
_i64 = ctypes.c_int64
_st = lambda: ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
_i64s = lambda t, dims: (ctypes.c_int64 * len(dims))(*[t.stride(d) for d in dims])


def _check(rc, what):
    if rc:
        raise RuntimeError(f"{what} failed ({rc})")        # the reference's callers see what a pybind exception gives them


class _PlanLayout(ctypes.Structure):                       # dpvo_plan_layout_t (include/dpvo_hip.h): 13 x int64
    _fields_ = [(n, ctypes.c_int64) for n in ("perm_k", "ku", "kx", "patch_off", "ix", "jx", "perm_p", "pu", "pair_off", "pair_ij",
                                              "counts", "total_ints", "flow")]


class cuda_corr:                                           # replaces: import cuda_corr        (correlation.cpp:57-63)
    @staticmethod
    def forward(fmap1, fmap2, coords, ii, jj, radius):     # correlation.cpp:58 -> dpvo_corr_forward
        B, N1, C, P, _ = fmap1.shape
        _, N2, _, H2, W2 = fmap2.shape
        M, D1 = coords.shape[1], 2 * radius + 1
        coords = coords.float().contiguous()
        ii, jj = ii.long().contiguous(), jj.long().contiguous()
        out = torch.empty(B, M, D1, D1, P, P, dtype=fmap1.dtype, device=fmap1.device)
        for b in range(B):
            _check(_L.dpvo_corr_forward(_p(fmap1[b]), _i64s(fmap1, (1, 2, 3, 4)), _p(fmap2[b]), _i64s(fmap2, (1, 2, 3, 4)),
                                        _p(coords[b]), ctypes.c_float(1.0), _p(ii), _p(jj), _p(out[b]),
                                        0 if fmap1.dtype == torch.float16 else 1, _i64(M), C, P, _i64(N1), _i64(N2), H2, W2, radius,
                                        _st()), "dpvo_corr_forward")
        return [out.permute(0, 1, 3, 2, 4, 5)]              # correlation_kernel.cu:232

    @staticmethod
    def patchify_forward(net, coords, radius):              # correlation.cpp:61 -> dpvo_patchify_forward
        B, C, H, W = net.shape
        M, D = coords.shape[1], 2 * radius + 2
        co = coords.float().contiguous()
        out = torch.empty(B, M, C, D, D, dtype=net.dtype, device=net.device)
        for b in range(B):
            _check(_L.dpvo_patchify_forward(_p(net[b]), _i64s(net, (1, 2, 3)), _p(co[b]), _p(out[b]),
                                            0 if net.dtype == torch.float16 else 1, _i64(M), C, H, W, radius, _st()),
                   "dpvo_patchify_forward")
        return [out]


class cuda_ba:                                             # replaces: import cuda_ba          (ba.cpp:183-189)
    @staticmethod
    def neighbors(kk, jj):                                  # ba.cpp:187 -> dpvo_neighbors (device resident, int64 out)
        kk, jj = kk.long().contiguous(), jj.long().contiguous()
        E = kk.numel()
        ix, jx = torch.empty_like(kk), torch.empty_like(kk)
        ws = torch.empty(max(1, _L.dpvo_neighbors_workspace_bytes(_i64(E))), dtype=torch.uint8, device=kk.device)
        _check(_L.dpvo_neighbors(_p(kk), _p(jj), _p(ix), _p(jx), _i64(E), _p(ws), ctypes.c_size_t(ws.numel()), _st()), "dpvo_neighbors")
        return [ix, jx]

    @staticmethod
    def forward(poses, patches, intrinsics, target, weight, lmbda, ii, jj, kk, PPF, t0, t1, iterations, eff_impl):
        """ba.cpp:184 -> dpvo_plan_build + dpvo_ba: updates poses / patches IN PLACE, returns [] (ba_cuda.cu:570-582).
        (the dense path, 6 (t1 - t0) <= 120; the block-sparse global BA is dpvo_gba_*: INTEGRATION.md §6)"""
        ii, jj, kk = ii.long().contiguous(), jj.long().contiguous(), kk.long().contiguous()
        E = ii.numel()
        target, weight = target.reshape(-1, 2).float().contiguous(), weight.reshape(-1, 2).float().contiguous()
        lay = _PlanLayout()
        _L.dpvo_plan_layout(_i64(E), ctypes.byref(lay))
        plan = torch.empty(lay.total_ints, dtype=torch.int32, device=ii.device)
        ws = torch.empty(max(_L.dpvo_plan_workspace_bytes(_i64(E)), _L.dpvo_ba_workspace_bytes(_i64(E), int(t1 - t0)), 1),
                         dtype=torch.uint8, device=ii.device)
        _check(_L.dpvo_plan_build(_p(ii), _p(jj), _p(kk), _i64(E), _p(plan), _p(ws), ctypes.c_size_t(ws.numel()), _st()), "dpvo_plan_build")
        # (dpvo_plan_build_ranged(..., n_frames=BUFFER_SIZE, n_patch_ids=BUFFER_SIZE*PPF, stream) sorts 32-bit keys instead)
        _check(_L.dpvo_ba(_p(poses), _p(patches), _p(intrinsics), _p(target), _p(weight), ctypes.c_float(float(lmbda)),
                          _p(ii), _p(jj), _p(kk), _p(plan), _i64(0), _i64(0), _i64(E), int(patches.shape[-1]), int(t0), int(t1),
                          int(iterations), None, _p(ws), ctypes.c_size_t(ws.numel()), _st()), "dpvo_ba")
        return []                                            # the caller's bare `except:` (dpvo.py:355) still applies

    @staticmethod
    def reproject(poses, patches, intrinsics, ii, jj, kk):  # ba.cpp:188 -> dpvo_reproject(clamp_z = 0: raw Z, ba_cuda.cu:422-423)
        ii, jj, kk = ii.long().contiguous(), jj.long().contiguous(), kk.long().contiguous()
        E, P = ii.numel(), patches.shape[-1]
        coords = torch.empty(1, E, 2, P, P, dtype=torch.float32, device=poses.device)
        _check(_L.dpvo_reproject(_p(poses), _p(patches), _p(intrinsics), _p(ii), _p(jj), _p(kk), _p(coords), _i64(E), int(P), 0, _st()),
               "dpvo_reproject")
        return coords

    @staticmethod
    def solve_system(J_Ginv_i, J_Ginv_j, ii, jj, res, ep, lm, freen):      # ba.cpp:188 -> dpvo_solve_system (f64 on the device; the reference: Eigen on the CPU)
        Ji, Jj, rs = J_Ginv_i.float().contiguous(), J_Ginv_j.float().contiguous(), res.reshape(-1, 7).float().contiguous()
        ii, jj = ii.long().contiguous(), jj.long().contiguous()
        r, n = ii.numel(), int(max(ii.max().item(), jj.max().item())) + 1
        delta = torch.empty(n, 7, dtype=torch.float32, device=rs.device)
        ws = torch.empty(_L.dpvo_solve_system_workspace_bytes(_i64(n), _i64(int(freen))), dtype=torch.uint8, device=rs.device)
        _check(_L.dpvo_solve_system(_p(Ji), _p(Jj), _p(ii), _p(jj), _p(rs), _i64(r), _i64(n), ctypes.c_float(float(ep)), ctypes.c_float(float(lm)),
                                    _i64(int(freen)), _p(delta), None, _p(ws), ctypes.c_size_t(ws.numel()), _st()), "dpvo_solve_system")
        return [delta]


class lietorch_backends:                                   # replaces: import lietorch_backends (lietorch.cpp:286-316), SE3 forward
    @staticmethod
    def _op(name, n_in, out_dim):
        def f(group_id, *xs):
            assert group_id == 3, "SE3 only"                 # groups.py:268-271
            xs = [x.float().contiguous() for x in xs[:n_in]]
            n = xs[0].shape[0]
            out = torch.empty(n, out_dim, dtype=torch.float32, device=xs[0].device)
            _check(getattr(_L, name)(*[_p(x) for x in xs], _p(out), _i64(n), _st()), name)
            return out
        return f


lietorch_backends.expm = lietorch_backends._op("dpvo_se3_exp", 1, 7)     # lietorch.cpp:288
lietorch_backends.logm = lietorch_backends._op("dpvo_se3_log", 1, 6)
lietorch_backends.inv = lietorch_backends._op("dpvo_se3_inv", 1, 7)
lietorch_backends.mul = lietorch_backends._op("dpvo_se3_mul", 2, 7)
lietorch_backends.act4 = lietorch_backends._op("dpvo_se3_act4", 2, 4)

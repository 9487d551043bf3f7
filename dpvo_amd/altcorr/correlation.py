"""altcorr: drop-in for the reference's `dpvo.altcorr` (dpvo/altcorr/correlation.py:51-74) on top of the HIP
kernels of dpvo_amd/csrc/corr.hip.  Inference only (the backward kernels are training-only in the reference).
"""
import torch

from .. import _lib as L

# bench.py sets this to a list to collect (start_event, end_event, n_edges) of every fused-correlation launch:
# HIP events recorded on the launch stream around the kernel (roofline measurement), None = no overhead.
PROFILE = None


def corr(fmap1, fmap2, coords, ii, jj, radius=1, dropout=1):
    """cuda_corr.forward (correlation.py:4-13,71-72).  fmap1 [B,N1,C,P,P], fmap2 [B,N2,C,H2,W2] (any strides,
    f16/f32), coords [B,M,2,P,P] f32, ii/jj [M] long -> [B, M, 2r+1 (x), 2r+1 (y), P, P] in the feature dtype."""
    L.require_cuda(fmap1, fmap2, coords, ii, jj)
    B, N1, C, P, _ = fmap1.shape
    _, N2, _, H2, W2 = fmap2.shape
    M = coords.shape[1]
    D1 = 2 * radius + 1
    coords = coords.float().contiguous()
    ii = ii.long().contiguous(); jj = jj.long().contiguous()
    if fmap2.dtype != fmap1.dtype:
        fmap2 = fmap2.to(fmap1.dtype)
    out = torch.empty(B, M, D1, D1, P, P, dtype=fmap1.dtype, device=fmap1.device)
    for b in range(B):
        L.check(L.lib().dpvo_corr_forward(
            L.ptr(fmap1[b]), L.strides(fmap1, (1, 2, 3, 4)), L.ptr(fmap2[b]), L.strides(fmap2, (1, 2, 3, 4)),
            L.ptr(coords[b]), L.f32(1.0), L.ptr(ii), L.ptr(jj), L.ptr(out[b]), L.i32(L.dtype_code(fmap1.dtype)),
            L.i64(M), L.i32(C), L.i32(P), L.i64(N1), L.i64(N2), L.i32(H2), L.i32(W2), L.i32(radius), L.stream()),
            "dpvo_corr_forward")
    return out.permute(0, 1, 3, 2, 4, 5)


def corr_pyramid(gmap_cl, fmap0_cl, fmap1_cl, coords, ii, jj, radius=3, order=None, out=None, ld_out=896):
    """Fused DPVO.corr (dpvo/dpvo.py:200-207): returns the [E, 882] f16 view of a [E, ld_out] buffer.

    gmap_cl [N1, P*P, C], fmap0_cl [N2, H0, W0, C], fmap1_cl [N2, H1, W1, C]: channels-last f16 storage."""
    L.require_cuda(gmap_cl, fmap0_cl, fmap1_cl, coords, ii, jj)
    assert gmap_cl.dtype == fmap0_cl.dtype == fmap1_cl.dtype == torch.float16
    assert gmap_cl.is_contiguous() and fmap0_cl.is_contiguous() and fmap1_cl.is_contiguous()
    N1, PP, C = gmap_cl.shape
    N2, H0, W0, _ = fmap0_cl.shape
    _, H1, W1, _ = fmap1_cl.shape
    P = int(round(PP ** 0.5))
    E = ii.numel()
    coords = coords.reshape(E, 2, P, P).float().contiguous()
    ii = ii.long().contiguous(); jj = jj.long().contiguous()
    nfeat = 2 * (2 * radius + 1) ** 2 * PP
    if out is None:
        out = torch.empty(E, ld_out, dtype=torch.float16, device=coords.device)
    prof = PROFILE
    if prof is not None:
        ev0 = torch.cuda.Event(enable_timing=True); ev1 = torch.cuda.Event(enable_timing=True)
        ev0.record()
    L.check(L.lib().dpvo_corr_pyramid_forward(
        L.ptr(gmap_cl), L.ptr(fmap0_cl), L.ptr(fmap1_cl), L.ptr(coords), L.ptr(ii), L.ptr(jj), L.ptr(order),
        L.ptr(out), L.i64(out.stride(0)), L.i64(E), L.i32(C), L.i32(P), L.i64(N1), L.i64(N2), L.i32(H0), L.i32(W0),
        L.i32(H1), L.i32(W1), L.i32(radius), L.stream()), "dpvo_corr_pyramid_forward")
    if prof is not None:
        ev1.record()
        prof.append((ev0, ev1, E))
    return out[:, :nfeat]


def patchify(net, coords, radius, mode='bilinear'):
    """altcorr.patchify (correlation.py:51-68): net [B,C,H,W], coords [B,M,2] -> [B,M,C,d,d]
    (d = 2r+1 with the bilinear blend fused in one kernel, d = 2r+2 for mode != 'bilinear')."""
    L.require_cuda(net, coords)
    B, C, H, W = net.shape
    M = coords.shape[1]
    co = coords.float().contiguous()
    bil = (mode == 'bilinear')
    D = 2 * radius + (1 if bil else 2)
    fn = L.lib().dpvo_patchify_bilinear if bil else L.lib().dpvo_patchify_forward
    patches = torch.empty(B, M, C, D, D, dtype=net.dtype, device=net.device)
    for b in range(B):
        L.check(fn(L.ptr(net[b]), L.strides(net, (1, 2, 3)), L.ptr(co[b]), L.ptr(patches[b]),
                   L.i32(L.dtype_code(net.dtype)), L.i64(M), L.i32(C), L.i32(H), L.i32(W), L.i32(radius), L.stream()),
                "dpvo_patchify")
    return patches

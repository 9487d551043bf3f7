"""Front-end helper kernels (dpvo_amd/csrc/frontend.hip) against the torch / lietorch formulation the reference uses."""
import ctypes

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from dpvo_amd import _lib as L
from dpvo_amd import altcorr, lietorch
from dpvo_amd.patchgraph import EdgeStore
from dpvo_amd import synthetic as S

pytestmark = pytest.mark.gpu


def test_normalize_colors_features(dev):
    g = torch.Generator().manual_seed(0)
    H, W, M = 96, 128, 40
    img = torch.randint(0, 256, (3, H, W), generator=g, dtype=torch.uint8).to(dev)
    f32 = torch.empty(3, H, W, device=dev); f16 = torch.empty(3, H, W, dtype=torch.float16, device=dev)
    L.check(L.lib().dpvo_normalize_image(L.ptr(img), L.ptr(f32), L.ptr(f16), L.i64(img.numel()), L.stream()), "norm")
    ref = 2 * (img[None, None] / 255.0) - 0.5                          # dpvo.py:389
    assert torch.equal(f32, ref[0, 0]) and torch.equal(f16, ref[0, 0].half())
    # sizes that are not a multiple of 16 and pointers that are not 16-byte aligned take the scalar tail / scalar path
    flat = img.reshape(-1)
    for off, n in ((0, 1000 + 7), (3, 4096), (16, 16 * 50 + 1)):
        a32 = torch.full((n + 8,), -9.0, device=dev); a16 = torch.full((n + 8,), -9.0, dtype=torch.float16, device=dev)
        src = flat[off:off + n]
        L.check(L.lib().dpvo_normalize_image(L.ptr(src), L.ptr(a32), L.ptr(a16), L.i64(n), L.stream()), "norm")
        r = 2 * (src / 255.0) - 0.5
        assert torch.equal(a32[:n], r) and torch.equal(a16[:n], r.half()) and (a32[n:] == -9).all() and (a16[n:] == -9).all()
        L.check(L.lib().dpvo_normalize_image(L.ptr(src), L.ptr(None), L.ptr(a16), L.i64(n), L.stream()), "norm")   # f16 only
        assert torch.equal(a16[:n], r.half())
    # colours: clr = patchify(images[0], 4*(coords+0.5), 0); clr = (clr[0,:,[2,1,0]] + 0.5) * (255/2) -> uint8
    coords = torch.stack([torch.randint(1, W // 4 - 1, (M,), generator=g), torch.randint(1, H // 4 - 1, (M,), generator=g)], -1).float().to(dev)
    coords[0] = torch.tensor([W / 4 - 0.3, 2.0], device=dev)           # partly out of bounds
    clr = altcorr.patchify(ref[0], 4 * (coords[None] + 0.5), 0).view(1, -1, 3)
    clr = ((clr[0, :, [2, 1, 0]] + 0.5) * (255.0 / 2)).to(torch.uint8)
    out = torch.zeros(M, 3, dtype=torch.uint8, device=dev)
    L.check(L.lib().dpvo_patch_colors(L.ptr(img), L.ptr(coords.contiguous()), L.ptr(out), L.i32(M), L.i32(H), L.i32(W), L.stream()), "clr")
    assert torch.equal(out, clr)
    # feature store: channels-last copy + 4x4 average pool
    for dt in (torch.float16, torch.float32):
        C, h, w = 128, 24, 72
        fm = torch.randn(C, h, w, generator=g).to(dt).to(dev)
        f1 = torch.zeros(h, w, C, dtype=dt, device=dev); f2 = torch.zeros(h // 4, w // 4, C, dtype=dt, device=dev)
        L.check(L.lib().dpvo_store_features(L.ptr(fm), L.ptr(f1), L.ptr(f2), L.i32(L.dtype_code(dt)), L.i32(C), L.i32(h), L.i32(w), L.stream()), "feat")
        assert torch.equal(f1, fm.permute(1, 2, 0))
        ref2 = F.avg_pool2d(fm[None].float(), 4, 4)[0].permute(1, 2, 0)
        assert torch.allclose(f2.float(), ref2, atol=2e-3 if dt == torch.float16 else 1e-6)


def test_motion_model_and_median(dev):
    g = torch.Generator().manual_seed(1)
    poses = lietorch.SE3.exp(0.3 * torch.randn(10, 6, generator=g).to(dev)).data.contiguous()
    poses[:, 3:] *= 1.3                                                # constructors normalise
    ref = poses.clone()
    n, scale = 7, 0.5 * 1.25
    P1, P2 = lietorch.SE3(ref[n - 1]), lietorch.SE3(ref[n - 2])        # dpvo.py:412-421
    xi = scale * (P1 * P2.inv()).log()
    ref[n] = (lietorch.SE3.exp(xi) * P1).data
    L.check(L.lib().dpvo_motion_model(L.ptr(poses), L.i32(n), L.f32(scale), L.stream()), "motion")
    assert torch.allclose(poses[n], ref[n], atol=2e-6) and torch.equal(poses[:n], ref[:n]) and torch.equal(poses[n + 1:], ref[n + 1:])
    for M in (16, 96):
        patches = torch.rand(12, M, 3, 3, 3, generator=g).to(dev)
        ref = patches.clone()
        n = 9
        ref[n, :, 2] = torch.median(ref[n - 3:n, :, 2])                # dpvo.py:430-432
        L.check(L.lib().dpvo_median_depth(L.ptr(patches), L.i32(n), L.i32(M), L.i32(3), L.stream()), "median")
        assert torch.equal(patches, ref)


def test_edge_store_matches_replay(dev):
    """append_frame + keep reproduce dpvo.py:215-238,362-375 exactly (compare with the CPU replay of the same rules)"""
    cfg = S.GraphCfg(M=8, REMOVAL_WINDOW=10, PATCH_LIFETIME=6)
    N = 64
    index_ = torch.zeros(N, cfg.M, dtype=torch.long, device=dev)
    st = EdgeStore(384, dev, cap=256)                                  # small capacity: exercises growth
    inac = EdgeStore(384, dev, with_state=False, cap=64)
    for n in range(1, 25):
        index_[n] = n
        E0 = st.E
        st.append_frame(index_.view(-1), n, cfg.M, cfg.PATCH_LIFETIME)
        assert (st.view("net")[E0:] == 0).all()
        st.view("net")[E0:] += n                                       # tag rows to check the gather moves them along
        st.view("target")[E0:] = float(n)
        m = st.view("ii") < n - cfg.REMOVAL_WINDOW
        rem = m.nonzero().squeeze(1)
        if rem.numel():
            inac.reserve(rem.numel()); st.gather_into(rem, inac.a, inac.E); inac.E += rem.numel()
        st.keep((~m).nonzero().squeeze(1))
        ii, jj, kk = S.replay_graph(n, cfg)
        assert torch.equal(st.view("ii").cpu(), ii) and torch.equal(st.view("jj").cpu(), jj) and torch.equal(st.view("kk").cpu(), kk)
        # the row tag equals the frame count at which the edge was created = jj + 1 for forward edges, kk//M + 1 for back
        born = torch.maximum(st.view("jj"), st.view("kk") // cfg.M) + 1
        assert torch.equal(st.view("net")[:, 0], born.float()) and torch.equal(st.view("target")[:, 1], born.float())
    assert inac.E > 0 and (inac.view("ii") < 24 - cfg.REMOVAL_WINDOW).all()


@pytest.mark.parametrize("mode", ["float_coords", "randint"])
def test_frame_patches_matches_patchifier_composition(dev, mode):
    """dpvo_frame_patches == the separate patchify gathers (net.py:136-147) + state stores (dpvo.py:401-438), bit for bit"""
    from dpvo_amd.utils import coords_grid_with_index
    g = torch.Generator().manual_seed(3)
    H, W, M, CF, CI = 96, 128, 48, 128, 384
    h, w = H // 4, W // 4
    img = torch.randint(0, 256, (3, H, W), generator=g, dtype=torch.uint8).to(dev)
    fmap = (torch.randn(h, w, CF, generator=g) / 2).half().to(dev)
    imap = (torch.randn(h, w, CI, generator=g) / 2).half().to(dev)
    xs = torch.randint(1, w - 1, (1, M), generator=g).to(dev)
    ys = torch.randint(1, h - 1, (1, M), generator=g).to(dev)
    coords = torch.stack([xs, ys], -1).float()
    if mode == "float_coords":
        coords = coords + torch.rand(1, M, 2, generator=g).to(dev) * 0.9
        coords[0, 0] = torch.tensor([w - 1.4, 0.3], device=dev)           # windows partly out of bounds
        coords[0, 1] = torch.tensor([0.2, h - 1.1], device=dev)
    depth = torch.rand(M, generator=g).to(dev)
    intr = torch.tensor([320.0, 321.0, 64.0, 48.0], device=dev)
    # composition (the code path DPVO uses without the fused kernel)
    f_nchw, i_nchw = fmap.permute(2, 0, 1)[None], imap.permute(2, 0, 1)[None]
    ref_imap = altcorr.patchify(i_nchw, coords, 0).view(M, CI)
    ref_gmap = altcorr.patchify(f_nchw, coords, 1).view(M, CF, 3, 3).permute(0, 2, 3, 1)
    grid, _ = coords_grid_with_index(torch.ones(1, 1, h, w, device=dev), device=dev)
    ref_patches = altcorr.patchify(grid[0], coords, 1).view(M, 3, 3, 3).float().clone()
    ref_patches[:, 2] = depth.view(M, 1, 1)
    img_n = 2 * (img[None, None] / 255.0) - 0.5
    clr = altcorr.patchify(img_n[0], 4 * (coords + 0.5), 0).view(1, -1, 3)
    ref_clr = ((clr[0, :, [2, 1, 0]] + 0.5) * (255.0 / 2)).to(torch.uint8)
    # fused
    gmap = torch.zeros(M, 3, 3, CF, dtype=torch.float16, device=dev); im = torch.zeros(M, CI, dtype=torch.float16, device=dev)
    patches = torch.zeros(M, 3, 3, 3, device=dev); colors = torch.zeros(M, 3, dtype=torch.uint8, device=dev)
    intr_o = torch.zeros(4, device=dev); idx_row = torch.zeros(M, dtype=torch.long, device=dev)
    idx_map = torch.zeros(1, dtype=torch.long, device=dev); cout = torch.zeros(M, 2, device=dev)
    use_xy = mode == "randint"
    L.check(L.lib().dpvo_frame_patches(
        L.ptr(fmap), L.ptr(imap), L.ptr(img), L.ptr(None if use_xy else coords[0].contiguous()),
        L.ptr(xs if use_xy else None), L.ptr(ys if use_xy else None), L.ptr(depth), L.ptr(intr), L.f32(4.0), L.ptr(gmap),
        L.ptr(im), L.ptr(patches), L.ptr(colors), L.ptr(intr_o), L.ptr(idx_row), L.ptr(idx_map), L.ptr(cout), L.i32(M),
        L.i32(h), L.i32(w), L.i32(H), L.i32(W), L.i32(CF), L.i32(CI), L.i32(3), L.i64(7), L.i64(7 * M), L.stream()),
        "dpvo_frame_patches")
    assert torch.equal(gmap, ref_gmap) and torch.equal(im, ref_imap)
    assert torch.equal(patches, ref_patches) and torch.equal(colors, ref_clr)
    assert torch.equal(intr_o, intr / 4.0) and torch.equal(cout, coords[0])
    assert (idx_row == 7).all() and idx_map.item() == 7 * M
    # the same as two launches (state stores first, feature gathers later): untouched groups stay untouched
    gmap2 = torch.zeros_like(gmap); im2 = torch.zeros_like(im); patches2 = torch.zeros_like(patches); colors2 = torch.zeros_like(colors)
    cc = L.ptr(None if use_xy else coords[0].contiguous())
    xy = (L.ptr(xs if use_xy else None), L.ptr(ys if use_xy else None))
    dims = (L.i32(M), L.i32(h), L.i32(w), L.i32(H), L.i32(W), L.i32(CF), L.i32(CI), L.i32(3), L.i64(7), L.i64(7 * M), L.stream())
    nul = L.ptr(None)
    L.check(L.lib().dpvo_frame_patches(nul, nul, L.ptr(img), cc, *xy, L.ptr(depth), L.ptr(intr), L.f32(4.0), nul, nul,
                                       L.ptr(patches2), L.ptr(colors2), nul, nul, nul, nul, *dims), "dpvo_frame_patches")
    assert torch.equal(patches2, ref_patches) and torch.equal(colors2, ref_clr) and not gmap2.any() and not im2.any()
    L.check(L.lib().dpvo_frame_patches(L.ptr(fmap), L.ptr(imap), nul, cc, *xy, nul, nul, L.f32(4.0), L.ptr(gmap2), L.ptr(im2),
                                       nul, nul, nul, nul, nul, nul, *dims), "dpvo_frame_patches")
    assert torch.equal(gmap2, ref_gmap) and torch.equal(im2, ref_imap)
    assert L.lib().dpvo_frame_patches(nul, nul, nul, cc, *xy, nul, nul, L.f32(4.0), nul, nul, nul, nul, nul, nul, nul, nul, *dims) < 0


def test_frame_state_composite_equals_separate_entries(dev):
    """dpvo_frame_state == dpvo_frame_patches + dpvo_motion_model + dpvo_median_depth + dpvo_pool4_nhwc + dpvo_append_edges"""
    import ctypes
    g = torch.Generator().manual_seed(11)
    H, W, M, CF, CI, n, r, D = 96, 128, 16, 128, 384, 9, 13, 384
    h, w = H // 4, W // 4
    img = torch.randint(0, 256, (3, H, W), generator=g, dtype=torch.uint8).to(dev)
    fmap = (torch.randn(h, w, CF, generator=g) / 2).half().to(dev)
    imap = (torch.randn(h, w, CI, generator=g) / 2).half().to(dev)
    xs = torch.randint(1, w - 1, (1, M), generator=g).to(dev); ys = torch.randint(1, h - 1, (1, M), generator=g).to(dev)
    depth = torch.rand(M, generator=g).to(dev)
    intr = torch.tensor([320.0, 321.0, 64.0, 48.0], device=dev)
    poses0 = torch.randn(16, 7, generator=g); poses0[:, 3:] /= poses0[:, 3:].norm(dim=1, keepdim=True)
    patches0 = torch.rand(16, M, 3, 3, 3, generator=g)
    ix = (torch.arange(16 * M) // M).to(dev)
    E0 = 37

    def run(composite):
        st = dict(gmap=torch.zeros(M, 3, 3, CF, dtype=torch.float16, device=dev), im=torch.zeros(M, CI, dtype=torch.float16, device=dev),
                  colors=torch.zeros(M, 3, dtype=torch.uint8, device=dev), intr_o=torch.zeros(4, device=dev),
                  idx_row=torch.zeros(M, dtype=torch.long, device=dev), idx_map=torch.zeros(1, dtype=torch.long, device=dev),
                  poses=poses0.clone().to(dev), patches=patches0.clone().to(dev), f2=torch.zeros(h // 4, w // 4, CF, dtype=torch.float16, device=dev),
                  ii=torch.zeros(4000, dtype=torch.long, device=dev), jj=torch.zeros(4000, dtype=torch.long, device=dev),
                  kk=torch.zeros(4000, dtype=torch.long, device=dev), net=torch.ones(4000, D, device=dev))
        if composite:
            fs = L.FrameState()
            dp = lambda t: t.data_ptr()
            fs.fmap, fs.imap, fs.img_u8, fs.xs, fs.ys, fs.depth, fs.intrinsics = dp(fmap), dp(imap), dp(img), dp(xs), dp(ys), dp(depth), dp(intr)
            fs.gmap_slot, fs.imap_slot, fs.patches_slot, fs.colors_slot = dp(st["gmap"]), dp(st["im"]), dp(st["patches"][n]), dp(st["colors"])
            fs.intrinsics_slot, fs.index_row, fs.index_map = dp(st["intr_o"]), dp(st["idx_row"]), dp(st["idx_map"])
            fs.poses, fs.mm_n, fs.mm_scale = dp(st["poses"]), n, 0.5
            fs.patches_all, fs.md_n, fs.fmap2_slot = dp(st["patches"]), n, dp(st["f2"])
            fs.ii, fs.jj, fs.kk, fs.net, fs.ix = dp(st["ii"]), dp(st["jj"]), dp(st["kk"]), dp(st["net"]), dp(ix)
            fs.frame_next, fs.m_next, fs.E0, fs.res = n + 1, (n + 1) * M, E0, 4.0
            fs.M, fs.h, fs.w, fs.H, fs.W, fs.CF, fs.CI, fs.P, fs.ap_n, fs.ap_r, fs.D = M, h, w, H, W, CF, CI, 3, n + 1, r, D
            L.check(L.lib().dpvo_frame_state(ctypes.byref(fs), L.stream()), "dpvo_frame_state")
            st["n_new"] = fs.n_new
        else:
            nul = L.ptr(None)
            L.check(L.lib().dpvo_frame_patches(
                L.ptr(fmap), L.ptr(imap), L.ptr(img), nul, L.ptr(xs), L.ptr(ys), L.ptr(depth), L.ptr(intr), L.f32(4.0), L.ptr(st["gmap"]),
                L.ptr(st["im"]), L.ptr(st["patches"][n]), L.ptr(st["colors"]), L.ptr(st["intr_o"]), L.ptr(st["idx_row"]), L.ptr(st["idx_map"]),
                nul, L.i32(M), L.i32(h), L.i32(w), L.i32(H), L.i32(W), L.i32(CF), L.i32(CI), L.i32(3), L.i64(n + 1), L.i64((n + 1) * M),
                L.stream()), "dpvo_frame_patches")
            L.check(L.lib().dpvo_motion_model(L.ptr(st["poses"]), L.i32(n), L.f32(0.5), L.stream()), "dpvo_motion_model")
            L.check(L.lib().dpvo_median_depth(L.ptr(st["patches"]), L.i32(n), L.i32(M), L.i32(3), L.stream()), "dpvo_median_depth")
            L.check(L.lib().dpvo_pool4_nhwc(L.ptr(fmap), L.ptr(st["f2"]), L.i32(h), L.i32(w), L.i32(CF), L.stream()), "dpvo_pool4_nhwc")
            cnt = ctypes.c_int64(0)
            L.check(L.lib().dpvo_append_edges(L.ptr(st["ii"]), L.ptr(st["jj"]), L.ptr(st["kk"]), L.ptr(st["net"]), L.ptr(ix), L.i64(E0),
                                              L.i32(n + 1), L.i32(M), L.i32(r), L.i32(D), ctypes.byref(cnt), L.stream()), "dpvo_append_edges")
            st["n_new"] = cnt.value
        torch.cuda.synchronize()
        return st

    a, b = run(False), run(True)
    assert a["n_new"] == b["n_new"] > 0
    for k in a:
        if k != "n_new":
            assert torch.equal(a[k], b[k]), k
    assert not torch.equal(a["poses"][n].cpu(), poses0[n]) and not torch.equal(a["patches"][n].cpu(), patches0[n])

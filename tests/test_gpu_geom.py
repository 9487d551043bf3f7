"""SE3 / reprojection / plan parity vs the oracle (float64).  Tolerances: coordinates are O(100) px computed in
f32 -> atol 2e-3 px; SE3 elements O(1) -> atol 2e-5; integer structures bit-exact."""
import numpy as np
import pytest
import torch

from dpvo_amd import fastba, lietorch, synthetic as S
from dpvo_amd import projective_ops as pops
from dpvo_amd.graph import GraphPlan
from tests import helpers as H

pytestmark = pytest.mark.gpu


def test_se3_ops(oracle, dev):
    g = torch.Generator().manual_seed(0)
    a = 0.3 * torch.randn(500, 6, generator=g)
    a[0] = 0; a[1, 3:] = 1e-8; a[2, 3:] = torch.tensor([3.1, 0.0, 0.0])
    X = lietorch.SE3.exp(a.to(dev))
    H.assert_close(X.data.cpu().numpy(), oracle.se3_exp(a.numpy()), 2e-5, 2e-5, "exp")
    H.assert_close(X.log().cpu().numpy(), oracle.se3_log(oracle.se3_exp(a.numpy())), 5e-5, 5e-5, "log")
    Y = lietorch.SE3.exp(0.5 * torch.randn(500, 6, generator=g).to(dev))
    H.assert_close(X.inv().data.cpu().numpy(), oracle.se3_inv(X.data.cpu().numpy()), 2e-5, 2e-5, "inv")
    H.assert_close((X * Y).data.cpu().numpy(), oracle.se3_mul(X.data.cpu().numpy(), Y.data.cpu().numpy()), 2e-5, 2e-5, "mul")
    p = torch.randn(500, 4, generator=g).to(dev)
    H.assert_close((X * p).cpu().numpy(), oracle.se3_act4(X.data.cpu().numpy(), p.cpu().numpy()), 2e-5, 2e-5, "act4")
    # lietorch's own identities (run_tests.py:16-28) in f32
    H.assert_close(lietorch.SE3.exp(a.to(dev)).log().cpu().numpy()[3:], a.numpy()[3:], 2e-4, 2e-4, "log(exp(a)) == a")
    I = (X * X.inv()).log()
    assert I.abs().max().item() < 1e-5
    # broadcasting + unnormalised quaternion input (SO3 constructor normalises, so3.h:35-37)
    Z = lietorch.SE3(X.data[None, :3] * torch.tensor([1, 1, 1, 2, 2, 2, 2.0], device=dev))
    pts = torch.randn(4, 1, 4, generator=g).to(dev)
    out = Z * pts
    assert out.shape == (4, 3, 4)
    ref = oracle.se3_act4(np.broadcast_to(Z.data.cpu().numpy(), (4, 3, 7)).copy(), np.broadcast_to(pts.cpu().numpy(), (4, 3, 4)).copy())
    H.assert_close(out.cpu().numpy(), ref, 2e-5, 2e-5, "broadcast act4")


def test_reproject_flow_points(oracle, dev):
    ii, jj, kk = S.replay_graph(40)
    poses, patches, intr = S.make_scene(40)
    sel = torch.randperm(ii.numel(), generator=torch.Generator().manual_seed(1))[:3000]
    ii, jj, kk = ii[sel], jj[sel], kk[sel]
    patches[5::7, 2] *= -1.0          # some points behind the camera -> exercises the Z clamp
    co = pops.transform_coords(poses.to(dev), patches.to(dev), intr.to(dev), ii.to(dev), jj.to(dev), kk.to(dev))
    ref = oracle.reproject(poses.numpy(), patches.numpy(), intr.numpy(), ii.numpy(), jj.numpy(), kk.numpy())
    assert co.shape == (1, 3000, 2, 3, 3)
    H.assert_close(co[0].cpu().numpy(), ref, 2e-3, 1e-5, "reproject")
    x1 = pops.transform(lietorch.SE3(poses[None].to(dev)), patches[None].to(dev), intr[None].to(dev), ii.to(dev), jj.to(dev), kk.to(dev))
    assert x1.shape == (1, 3000, 3, 3, 2) and torch.equal(x1.permute(0, 1, 4, 2, 3), co)
    fl, val = pops.flow_mag(poses.to(dev), patches.to(dev), intr.to(dev), ii.to(dev), jj.to(dev), kk.to(dev), beta=0.5)
    rf, rv = oracle.flow_mag(poses.numpy(), patches.numpy(), intr.numpy(), ii.numpy(), jj.numpy(), kk.numpy(), beta=0.5)
    H.assert_close(fl.cpu().numpy(), rf.reshape(3000, -1).mean(1), 2e-3, 1e-4, "flow_mag")
    assert np.array_equal(val.cpu().numpy(), rv.reshape(3000, -1).sum(1))
    # fused keyframe test (DPVO.motionmag(i,j) + motionmag(j,i))
    ii_f, jj_f, kk_f = S.replay_graph(40)
    a, b = pops.motionmag_pair(poses.to(dev), patches.to(dev), intr.to(dev), ii_f.to(dev), jj_f.to(dev), kk_f.to(dev), 35, 37, beta=0.5)
    plan_f = GraphPlan(ii_f.to(dev), jj_f.to(dev), kk_f.to(dev))
    a2, b2 = pops.motionmag_pair(poses.to(dev), patches.to(dev), intr.to(dev), ii_f.to(dev), jj_f.to(dev), kk_f.to(dev), 35, 37, beta=0.5, plan=plan_f)
    assert abs(a - a2) < 1e-4 * max(1, abs(a)) and abs(b - b2) < 1e-4 * max(1, abs(b))
    a3, b3 = pops.motionmag_pair(poses.to(dev), patches.to(dev), intr.to(dev), ii_f.to(dev), jj_f.to(dev), kk_f.to(dev), 3, 39, plan=plan_f)
    assert a3 != a3 and b3 != b3
    for (qi, qj, got) in ((35, 37, a), (37, 35, b)):
        msk = (ii_f == qi) & (jj_f == qj)
        rf2, _ = oracle.flow_mag(poses.numpy(), patches.numpy(), intr.numpy(), ii_f[msk].numpy(), jj_f[msk].numpy(), kk_f[msk].numpy(), beta=0.5)
        assert msk.sum() == 96 and abs(got - rf2.mean()) < 2e-3 * max(1.0, abs(rf2.mean()))
    a, b = pops.motionmag_pair(poses.to(dev), patches.to(dev), intr.to(dev), ii_f.to(dev), jj_f.to(dev), kk_f.to(dev), 3, 39)
    assert a != a and b != b          # no such edges: NaN like torch's mean of an empty tensor
    m = 40 * 96
    ix = torch.arange(m) // 96
    pts = pops.point_cloud(poses.to(dev), patches[:m].to(dev), intr.to(dev), ix.to(dev))
    rp = oracle.point_cloud(poses.numpy(), patches[:m].numpy(), intr.numpy(), ix.numpy())
    H.assert_close(pts.cpu().numpy(), rp, 1e-3, 1e-4, "point_cloud")
    # exported-but-unused cuda_ba.reproject semantics (raw Z, intrinsics[0])
    co2 = fastba.reproject(poses.to(dev), patches.to(dev), intr.to(dev), ii.to(dev), jj.to(dev), kk.to(dev))
    ok = torch.from_numpy(ref[:, 0, 1, 1]).abs() < 1e4
    front = (patches[kk, 2, 1, 1] > 0)
    H.assert_close(co2[0].cpu().numpy()[(ok & front).numpy()], ref[(ok & front).numpy()], 5e-3, 1e-4, "cuda_ba.reproject")


@pytest.mark.parametrize("ranged", [False, True, "window", "wide"])
@pytest.mark.parametrize("case", ["replay40", "small", "shuffled", "single", "empty", "tiles"])
def test_plan_bit_exact(oracle, dev, case, ranged):
    if case == "replay40":
        ii, jj, kk = S.replay_graph(40)
    elif case == "small":
        ii, jj, kk, _ = H.small_graph()
    elif case == "shuffled":
        ii, jj, kk = S.replay_graph(30)
        p = torch.randperm(ii.numel(), generator=torch.Generator().manual_seed(2))
        ii, jj, kk = ii[p], jj[p], kk[p]
        # duplicate edges (same patch, same frame) must keep edge order (stable sort, ba.cpp:80-82)
        ii = torch.cat([ii, ii[:500]]); jj = torch.cat([jj, jj[:500]]); kk = torch.cat([kk, kk[:500]])
    elif case == "single":
        ii, jj, kk = torch.tensor([3]), torch.tensor([5]), torch.tensor([300])
    elif case == "tiles":               # exactly two 1024-edge tiles, shuffled, with duplicates
        ii, jj, kk = S.replay_graph(12)
        p = torch.randperm(ii.numel(), generator=torch.Generator().manual_seed(5))[:2048 - 300]
        ii, jj, kk = ii[p], jj[p], kk[p]
        ii = torch.cat([ii, ii[:300]]); jj = torch.cat([jj, jj[:300]]); kk = torch.cat([kk, kk[:300]])
    else:
        ii = jj = kk = torch.zeros(0, dtype=torch.long)
    E = ii.numel()
    # ranged: bounds on the index values -> 32-bit keys, partial-width radix sorts (dpvo_plan_build_ranged)
    # window: all ids inside small windows -> counting-sort build (dpvo_plan_build_window), same plan bit for bit
    rng = dict(n_frames=4096, n_patch_ids=4096 * 96) if ranged else {}
    if ranged == "window" and E:
        flo = int(min(ii.min(), jj.min())); nfw = int(max(ii.max(), jj.max())) + 1 - flo
        plo = int(kk.min()); npw = int(kk.max()) + 1 - plo
        assert nfw * nfw <= 2048 and npw <= 4096, "test graphs are meant to fit the window path"
        rng["window"] = (flo, nfw, plo, npw)
    if ranged == "wide" and E:
        # wide: ids bounded by the frame count only -> bins-in-memory counting build (dpvo_plan_build_wide), same plan bit for bit
        rng["wide"] = (int(max(ii.max(), jj.max())) + 1, int(kk.max()) + 1)
    plan = GraphPlan(ii.to(dev), jj.to(dev), kk.to(dev), **rng)
    if ranged == "window" and E:
        assert plan.counts.cpu().tolist()[3] == 0
    if ranged == "wide" and E:
        assert plan.wide and plan.counts.cpu().tolist()[2:] == [0, 0]
        radix = GraphPlan(ii.to(dev), jj.to(dev), kk.to(dev))
        ng, npair = radix.n_patches(), radix.n_pairs()
        assert plan.counts.cpu().tolist()[:2] == [ng, npair]
        for name, cnt in (("perm_k", E), ("ku", E), ("kx", ng), ("patch_off", ng + 1), ("ix", E), ("jx", E), ("perm_p", E), ("pu", E),
                          ("pair_off", npair + 1), ("pair_ij", 2 * npair)):
            assert torch.equal(getattr(plan, name)[:cnt], getattr(radix, name)[:cnt]), name
        assert plan.flow.cpu().tolist()[:4] == radix.flow.cpu().tolist()[:4] == [-1, -1, 0, 0]
    if E == 0:
        assert plan.n_patches() == 0 and plan.n_pairs() == 0
        return
    ix, jx = oracle.neighbors(kk.numpy(), jj.numpy())
    assert np.array_equal(plan.ix.cpu().numpy()[:E], ix) and np.array_equal(plan.jx.cpu().numpy()[:E], jx)
    ix2, jx2 = fastba.neighbors(kk.to(dev), jj.to(dev))
    assert ix2.dtype == torch.long and np.array_equal(ix2.cpu().numpy(), ix) and np.array_equal(jx2.cpu().numpy(), jx)
    kx, ku = oracle.unique(kk.numpy())
    assert plan.n_patches() == kx.size
    assert np.array_equal(plan.kx.cpu().numpy()[:kx.size], kx) and np.array_equal(plan.ku.cpu().numpy()[:E], ku)
    _, pu = np.unique((ii * 12345 + jj).numpy(), return_inverse=True)      # net.py:88 / blocks.py:41
    assert np.array_equal(plan.pu.cpu().numpy()[:E], pu)
    assert plan.n_pairs() == pu.max() + 1
    # CSR consistency
    perm_k = plan.perm_k.cpu().numpy()[:E]; off = plan.patch_off.cpu().numpy()[:kx.size + 1]
    assert off[0] == 0 and off[-1] == E and sorted(perm_k.tolist()) == list(range(E))
    for g in (0, kx.size // 2, kx.size - 1):
        mem = perm_k[off[g]:off[g + 1]]
        assert (kk.numpy()[mem] == kx[g]).all() and (np.diff(jj.numpy()[mem]) >= 0).all()
    perm_p = plan.perm_p.cpu().numpy()[:E]; poff = plan.pair_off.cpu().numpy()[:plan.n_pairs() + 1]
    pij = plan.pair_ij.cpu().numpy()[:2 * plan.n_pairs()].reshape(-1, 2)
    for g in (0, plan.n_pairs() - 1):
        mem = perm_p[poff[g]:poff[g + 1]]
        assert (ii.numpy()[mem] == pij[g, 0]).all() and (jj.numpy()[mem] == pij[g, 1]).all()


def test_plan_window_reports_ids_outside_the_window(dev):
    """dpvo_plan_build_window: ids outside the promised window are clamped (no out-of-bounds access) and flagged in counts[3];
    windows too wide for the counting sort fall back to the radix build by themselves"""
    ii, jj, kk = S.replay_graph(20)
    E = ii.numel()
    flo, nfw, plo, npw = int(min(ii.min(), jj.min())), int(max(ii.max(), jj.max())) + 1, int(kk.min()), int(kk.max()) + 1
    good = GraphPlan(ii.to(dev), jj.to(dev), kk.to(dev), window=(flo, nfw - flo, plo, npw - plo))
    assert good.counts.cpu().tolist()[3] == 0
    bad = GraphPlan(ii.to(dev), jj.to(dev), kk.to(dev), window=(flo + 2, nfw - flo - 2, plo + 50, npw - plo - 50))
    torch.cuda.synchronize()
    assert bad.counts.cpu().tolist()[3] == 1
    assert int(bad.perm_k.min()) >= 0 and int(bad.perm_k.max()) < E and int(bad.perm_p.min()) >= 0 and int(bad.perm_p.max()) < E
    # the flag reaches the host with the frame's only read-back (the flow test of DPVO.keyframe): dpvo_motionmag_status
    from dpvo_amd import projective_ops as pops
    poses, patches, intr = (t.to(dev) for t in S.make_scene(20))
    for plan, flag in ((good, 0), (bad, 1)):
        host = torch.empty(8, dtype=torch.float32).pin_memory()
        fin = pops.motionmag_pair(poses, patches, intr, ii.to(dev), jj.to(dev), kk.to(dev), 14, 16, plan=plan, defer=True, host_buf=host)
        a, b = fin()
        assert fin.plan_status[3] == flag and fin.plan_status[2] == 0
        if flag == 0:
            assert fin.plan_status[:2] == (good.n_patches(), good.n_pairs()) and a == a and b == b
    wide = GraphPlan(ii.to(dev), jj.to(dev), kk.to(dev), window=(0, 4096, 0, 4096 * 96))        # -> dpvo_plan_build_ranged
    for name in ("perm_k", "ku", "ix", "jx", "perm_p", "pu"):
        assert torch.equal(getattr(wide, name), getattr(good, name)), name
    assert wide.counts.cpu().tolist()[:2] == good.counts.cpu().tolist()[:2]


def test_plan_wide_at_global_ba_size_and_outside_ids(dev):
    """dpvo_plan_build_wide at the size of the global BA's plan (active + inactive edges of a 110-frame run with loop-closure edges,
    duplicates included): every array equal to the radix build's; ids outside the promised ranges are clamped (no out-of-bounds
    access) and flagged in counts[3]; ranges it does not take fall back to the radix build by themselves"""
    ii, jj, kk = S.replay_graph(110)
    g = torch.Generator().manual_seed(11)
    # loop-closure edges: all patches of 40 old frames against recent frames, some of them twice (re-added in a later round)
    src = torch.randint(0, 60, (40,), generator=g); dst = torch.randint(95, 110, (40,), generator=g)
    lk = (src[:, None] * 96 + torch.arange(96)[None]).reshape(-1); lj = dst[:, None].expand(-1, 96).reshape(-1)
    ii = torch.cat([ii, lk // 96, lk[:960] // 96]); jj = torch.cat([jj, lj, lj[:960]]); kk = torch.cat([kk, lk, lk[:960]])
    p = torch.randperm(ii.numel(), generator=g)
    ii, jj, kk = ii[p], jj[p], kk[p]
    # ... with a stretch in append order (runs of equal bins inside a wave take the one-atomic-per-run path)
    a, b, c = S.replay_graph(20)
    ii = torch.cat([ii, a]); jj = torch.cat([jj, b]); kk = torch.cat([kk, c])
    E = ii.numel()
    d = lambda t: t.to(dev)
    radix = GraphPlan(d(ii), d(jj), d(kk), n_frames=4096, n_patch_ids=4096 * 96)
    wide = GraphPlan(d(ii), d(jj), d(kk), n_patches_ub=E, n_pairs_ub=E, wide=(111, 111 * 96))
    assert wide.wide and not radix.wide
    ng, npair = radix.n_patches(), radix.n_pairs()
    assert wide.counts.cpu().tolist() == [ng, npair, 0, 0]
    for name, cnt in (("perm_k", E), ("ku", E), ("kx", ng), ("patch_off", ng + 1), ("ix", E), ("jx", E), ("perm_p", E), ("pu", E),
                      ("pair_off", npair + 1), ("pair_ij", 2 * npair)):
        assert torch.equal(getattr(wide, name)[:cnt], getattr(radix, name)[:cnt]), name
    again = GraphPlan(d(ii), d(jj), d(kk), n_patches_ub=E, n_pairs_ub=E, wide=(111, 111 * 96))
    for name, cnt in (("perm_k", E), ("ix", E), ("jx", E), ("perm_p", E)):
        assert torch.equal(getattr(again, name)[:cnt], getattr(wide, name)[:cnt]), "the wide build must not depend on atomic order"
    bad = GraphPlan(d(ii), d(jj), d(kk), n_patches_ub=E, n_pairs_ub=E, wide=(100, 100 * 96))
    torch.cuda.synchronize()
    assert bad.wide and bad.counts.cpu().tolist()[3] == 1
    assert int(bad.perm_k.min()) >= 0 and int(bad.perm_k.max()) < E and int(bad.perm_p.min()) >= 0 and int(bad.perm_p.max()) < E
    far = GraphPlan(d(ii), d(jj), d(kk), wide=(4096, 4096 * 96))            # 4096^2 pair bins: -> dpvo_plan_build_ranged
    assert not far.wide
    for name in ("perm_k", "ku", "ix", "jx", "perm_p", "pu"):
        assert torch.equal(getattr(far, name), getattr(radix, name)), name


def test_plan_flow_list_and_flow_test_bits(dev):
    """dpvo_plan_build_window_flow: the plan's `flow` region lists the edges of ONE frame pair in both directions, in edge order; the
    flow test (DPVO.motionmag, dpvo.py:257-270) started from that list gives the same BITS as the walk through pair_ij / pair_off /
    perm_p; a pair without edges and every other builder leave the readers on the walk"""
    from dpvo_amd import projective_ops as pops
    ii, jj, kk = S.replay_graph(20)
    poses, patches, intr = (t.to(dev) for t in S.make_scene(20))
    flo, nfw, plo, npw = int(min(ii.min(), jj.min())), int(max(ii.max(), jj.max())) + 1, int(kk.min()), int(kk.max()) + 1
    win = (flo, nfw - flo, plo, npw - plo)
    d = lambda t: t.to(dev)
    plain = GraphPlan(d(ii), d(jj), d(kk), window=win)
    assert plain.flow.cpu().tolist()[:4] == [-1, -1, 0, 0]
    ranged = GraphPlan(d(ii), d(jj), d(kk), n_frames=4096, n_patch_ids=4096 * 96)
    assert ranged.flow.cpu().tolist()[:2] == [-1, -1]
    for qi, qj in ((14, 16), (16, 14), (3, 19)):
        listed = GraphPlan(d(ii), d(jj), d(kk), window=win, flow_pair=(qi, qj))
        fl = listed.flow.cpu().numpy()
        fwd, bwd = kk[(ii == qi) & (jj == qj)].numpy(), kk[(ii == qj) & (jj == qi)].numpy()
        assert fl[:4].tolist() == [qi, qj, fwd.size, bwd.size]
        assert np.array_equal(fl[4:4 + fwd.size], fwd) and np.array_equal(fl[4 + 256:4 + 256 + bwd.size], bwd)
        for name in ("perm_k", "ku", "ix", "jx", "perm_p", "pu", "counts"):
            assert torch.equal(getattr(listed, name), getattr(plain, name)), name
        ng = plain.n_pairs()
        assert torch.equal(listed.pair_off[:ng + 1], plain.pair_off[:ng + 1]) and torch.equal(listed.pair_ij[:2 * ng], plain.pair_ij[:2 * ng])
        outs = []
        for plan in (plain, listed, ranged):
            host = torch.empty(8, dtype=torch.float32).pin_memory()
            fin = pops.motionmag_pair(poses, patches, intr, d(ii), d(jj), d(kk), qi, qj, plan=plan, defer=True, host_buf=host)
            fin()
            outs.append(host[:4].clone())
        assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2]), (qi, qj, outs)
        assert (outs[0][1] == fwd.size) and (outs[0][3] == bwd.size)

"""End-to-end front-end test: dpvo_amd.dpvo.DPVO on a synthetic stream.

Integer bookkeeping (edge lists, frame / patch counters, timestamps, inactive-edge store) must match the numpy
restatement oracle/graph_ref.py BIT FOR BIT when both are driven through the same accept / keyframe decisions.
Floating-point state is checked for sanity (finite, unit quaternions, positive depths) and determinism."""
import numpy as np
import pytest
import torch

from dpvo_amd import projective_ops as pops
from dpvo_amd.config import cfg as base_cfg, DEFAULT_YAML
from dpvo_amd.dpvo import DPVO
from dpvo_amd.net import VONet

pytestmark = pytest.mark.gpu


def _run(dev, decisions, M=16, seed=0, ht=96, wd=128, defer=False, check=True, overlap=False, frame_call=True, host_images=None):
    """frame_call: steady-state frames through dpvo_frame_update + the device-side keyframe step (default), or the
    Python-paced round-2 path with its host mirror (dpvo_amd.dpvo._FRAME_CALL = False)"""
    from oracle.graph_ref import GraphRef
    import dpvo_amd.dpvo as dpvo_mod
    fc_before = dpvo_mod._FRAME_CALL
    dpvo_mod._FRAME_CALL = frame_call
    cfg = base_cfg.clone(); cfg.merge_from_dict(DEFAULT_YAML)
    cfg.PATCHES_PER_FRAME = M
    cfg.BUFFER_SIZE = 256
    torch.manual_seed(seed)
    slam = DPVO(cfg, VONet(), ht=ht, wd=wd, device=dev, defer_keyframe=defer, overlap_encoders=overlap)
    ref = GraphRef(M=M, PATCH_LIFETIME=cfg.PATCH_LIFETIME, REMOVAL_WINDOW=cfg.REMOVAL_WINDOW, BUFFER_SIZE=256)
    g = torch.Generator().manual_seed(seed)
    intr = torch.tensor([100.0, 100.0, wd / 2, ht / 2], device=dev)
    state = {}
    slam.motion_probe = lambda: 1e9 if state["accept"] else 0.0
    orig = pops.motionmag_pair
    thresh = cfg.KEYFRAME_THRESH
    calls = []

    def fake(*a, defer=False, host_buf=None, **k):
        calls.append(orig(*a, **k))           # the real kernel still runs (and must not crash)
        res = (0.0, 0.0) if state["drop"] else (4 * thresh, 4 * thresh)
        return (lambda: res) if defer else res
    pops.motionmag_pair = fake
    slam.keyframe_override = lambda counter: state["drop"]        # (the one-call path takes the scripted decision this way)
    try:
        for t, (accept, drop) in enumerate(decisions):
            state["accept"], state["drop"] = accept, drop
            img = torch.randint(0, 255, (3, ht, wd), generator=g, dtype=torch.uint8)
            if host_images is None:
                img = img.to(dev)
            elif host_images == "hwc":            # the reader's HWC buffer seen as CHW (demo.py:38 without its .cuda())
                img = img.permute(1, 2, 0).contiguous().permute(2, 0, 1)
            elif host_images == "numpy":
                img = img.numpy()
            slam(float(t), img, intr)
            ev = ref.frame(accept, drop)
            if not check:
                continue
            slam.flush()                          # (a deferred keyframe decision must be applied before state is read)
            assert slam.n == ref.n and slam.m == ref.m and slam.counter == ref.counter, (t, ev)
            assert np.array_equal(slam.pg.ii.cpu().numpy(), ref.ii), (t, ev)
            assert np.array_equal(slam.pg.jj.cpu().numpy(), ref.jj), (t, ev)
            assert np.array_equal(slam.pg.kk.cpu().numpy(), ref.kk), (t, ev)
            assert np.array_equal(slam.pg.kk_inac.cpu().numpy(), ref.kk_inac) and np.array_equal(slam.pg.jj_inac.cpu().numpy(), ref.jj_inac)
            assert np.array_equal(slam.pg.tstamps_[:slam.n], ref.tstamps_[:ref.n])
            assert slam.pg.net.shape == (1, ref.ii.size, 384) and slam.pg.net.dtype == torch.float32
    finally:
        pops.motionmag_pair = orig
        dpvo_mod._FRAME_CALL = fc_before
    return slam, ref, calls


@pytest.mark.parametrize("frame_call", [True, False])
def test_bookkeeping_bit_exact_and_state_sane(dev, frame_call):
    # skip two frames before init, then track; drop some keyframes, keep others
    decisions = [(True, False)] * 3 + [(False, False)] * 2 + [(True, False)] * 9 + [(True, True)] * 3 + \
                [(True, False)] * 22 + [(True, True), (True, False), (True, True)] + [(True, False)] * 4
    slam, ref, calls = _run(dev, decisions, frame_call=frame_call)
    assert slam.is_initialized and (frame_call or len(calls) > 20)
    n = slam.n
    P = slam.pg.poses_[:n]
    assert torch.isfinite(P).all()
    assert (P[:, 3:].norm(dim=-1) - 1).abs().max() < 1e-3
    d = slam.pg.patches_[:n, :, 2]
    assert torch.isfinite(d).all() and (d > 0).all()
    assert torch.isfinite(slam.pg.points_[:slam.m]).all()
    poses, tstamps = slam.terminate()
    assert poses.shape == (len(decisions), 7) and tstamps.shape == (len(decisions),)
    assert np.isfinite(poses).all()
    # every frame that was skipped / dropped is recoverable through pg.delta
    assert set(ref.delta.keys()) == set(int(k) for k in slam.pg.delta.keys())


def test_deferred_keyframe_is_bit_identical(dev):
    """defer_keyframe=True resolves each keyframe decision during the next frame's encoders: same state, bit for bit"""
    decisions = [(True, False)] * 12 + [(True, True)] * 2 + [(True, False)] * 20 + [(True, True), (True, False), (True, True)] + \
                [(True, False)] * 5
    a, ra, _ = _run(dev, decisions, seed=5)
    b, rb, _ = _run(dev, decisions, seed=5, defer=True, check=False)
    b.flush()
    c, _, _ = _run(dev, decisions, seed=5, defer=True, check=False, overlap=True)     # + encoders on a second stream
    c.flush()
    torch.cuda.synchronize()
    assert c.n == a.n and torch.equal(a.pg.poses_[:a.n], c.pg.poses_[:c.n]) and torch.equal(a.pg.patches_[:a.n], c.pg.patches_[:c.n])
    assert torch.equal(a.pg.net, c.pg.net) and torch.equal(a._fmap1_cl, c._fmap1_cl) and torch.equal(a._fmap2_cl, c._fmap2_cl)
    assert a.n == b.n == ra.n and a.m == b.m
    for k in ("ii", "jj", "kk"):
        assert torch.equal(getattr(a.pg, k), getattr(b.pg, k)) and np.array_equal(getattr(b.pg, k).cpu().numpy(), getattr(rb, k))
    assert torch.equal(a.pg.poses_[:a.n], b.pg.poses_[:b.n]) and torch.equal(a.pg.patches_[:a.n], b.pg.patches_[:b.n])
    assert torch.equal(a.pg.net, b.pg.net) and torch.equal(a._fmap1_cl, b._fmap1_cl) and torch.equal(a._gmap_cl, b._gmap_cl)
    pa, _ = a.terminate(); pb, _ = b.terminate()
    assert np.array_equal(pa, pb)


@pytest.mark.parametrize("overlap", [True, False])
def test_images_handed_over_in_host_memory(dev, overlap):
    """slam(t, image, intrinsics) with the image still in host memory -- a CHW tensor, the reader's HWC buffer behind a CHW view, a numpy
    array -- is uploaded by the tracker through its pinned ring on the encoder stream (DPVO._upload_image): the same tracker state, bit
    for bit, as with the image uploaded by the caller, with keyframes dropped along the way and the three ring slots reused many times."""
    decisions = [(True, False)] * 12 + [(True, True)] * 2 + [(True, False)] * 14 + [(True, True), (True, False)] + [(True, False)] * 4
    kw = dict(seed=11, defer=overlap, check=False, overlap=overlap)
    a, _, _ = _run(dev, decisions, **kw)
    a.flush(); torch.cuda.synchronize()
    for mode in ("chw", "hwc", "numpy"):
        b, _, _ = _run(dev, decisions, host_images=mode, **kw)
        b.flush(); torch.cuda.synchronize()
        assert b._img_ring is not None and a._img_ring is None
        assert a.n == b.n and a.pg.edges.E == b.pg.edges.E
        for k in ("ii", "jj", "kk", "net", "target", "weight"):
            assert torch.equal(getattr(a.pg, k), getattr(b.pg, k)), (mode, k)
        for k in ("poses_", "patches_", "colors_"):
            assert torch.equal(getattr(a.pg, k)[:a.n], getattr(b.pg, k)[:b.n]), (mode, k)
        assert torch.equal(a._fmap1_cl, b._fmap1_cl) and torch.equal(a.imap_, b.imap_), mode


def test_device_keyframe_step_equals_the_host_path(dev):
    """dpvo_frame_update + dpvo_keyframe_step (decision, removal of the dropped keyframe's edges, renumbering, ring shifts and
    the window removal all on the device, one C call per frame) against the Python-paced path with its host-side masks: the
    same tracker state bit for bit -- edges (active and inactive), hidden state, targets / weights, poses, depths, feature rings,
    the final trajectory -- in every mode (immediate, deferred, overlapped).  The relative pose stored for a dropped frame
    (dpvo.py:276) is computed inside dpvo_keyframe_step instead of by two lietorch launches: same formulas, but the compiler
    contracts multiply-adds per translation unit, so it agrees to f32 rounding (1e-6) with the host path and bit for bit
    between the modes of the device path."""
    decisions = [(True, False)] * 12 + [(True, True)] * 2 + [(True, False)] * 20 + [(True, True), (True, False), (True, True)] + \
                [(True, True)] * 3 + [(True, False)] * 6
    ref_run, rr, _ = _run(dev, decisions, seed=7, frame_call=False)
    first = None
    for kw in (dict(), dict(defer=True, check=False), dict(defer=True, check=False, overlap=True)):
        a, _, _ = _run(dev, decisions, seed=7, frame_call=True, **kw)
        a.flush()
        torch.cuda.synchronize()
        b = ref_run
        assert a.n == b.n == rr.n and a.m == b.m and a.pg.edges.E == b.pg.edges.E and a.pg.edges_inac.E == b.pg.edges_inac.E
        for k in ("ii", "jj", "kk", "ii_inac", "jj_inac", "kk_inac", "net", "target", "weight", "target_inac", "weight_inac"):
            assert torch.equal(getattr(a.pg, k), getattr(b.pg, k)), (kw, k)
        n = a.n
        for k in ("poses_", "patches_", "intrinsics_", "colors_"):
            assert torch.equal(getattr(a.pg, k)[:n], getattr(b.pg, k)[:n]), (kw, k)
        assert np.array_equal(a.pg.tstamps_[:n], b.pg.tstamps_[:n])
        for k in ("_fmap1_cl", "_fmap2_cl", "_gmap_cl", "imap_"):
            assert torch.equal(getattr(a, k), getattr(b, k)), (kw, k)
        assert set(a.pg.delta.keys()) == set(b.pg.delta.keys())
        for t in a.pg.delta:
            assert a.pg.delta[t][0] == b.pg.delta[t][0], (kw, t)
            assert (a.pg.delta[t][1].data - b.pg.delta[t][1].data).abs().max().item() < 1e-6, (kw, t)
            if first is not None:
                assert torch.equal(a.pg.delta[t][1].data, first.pg.delta[t][1].data), (kw, t)
        first = first or a
    pa, ta = a.terminate()
    pf, _ = first.terminate()
    pb, tb = ref_run.terminate()
    assert np.array_equal(pa, pf) and np.array_equal(ta, tb)
    assert np.abs(pa - pb).max() < 1e-5


def _run_unforced(dev, thresh, n_frames=44, M=16, ht=96, wd=128, seed=11, frame_call=True, **kw):
    """a tracker whose keyframe test is NOT scripted (keyframe_override None, forced = -1): the decision comes from the flow test"""
    import dpvo_amd.dpvo as dpvo_mod
    fc_before = dpvo_mod._FRAME_CALL
    dpvo_mod._FRAME_CALL = frame_call
    try:
        cfg = base_cfg.clone(); cfg.merge_from_dict(DEFAULT_YAML)
        cfg.PATCHES_PER_FRAME, cfg.BUFFER_SIZE, cfg.KEYFRAME_THRESH = M, 256, thresh
        torch.manual_seed(seed)
        slam = DPVO(cfg, VONet(), ht=ht, wd=wd, device=dev, **kw)
        slam.motion_probe = lambda: 1e9
        g = torch.Generator().manual_seed(seed)
        tex = torch.rand(3, ht + 64, wd + 64, generator=g)
        tex = torch.nn.functional.avg_pool2d(tex[None], 5, 1, 2)[0]
        tex = (255 * (tex - tex.min()) / (tex.max() - tex.min())).to(torch.uint8)
        intr = torch.tensor([100.0, 100.0, wd / 2, ht / 2], device=dev)
        log = []
        for t in range(n_frames):
            img = tex[:, (2 * t) % 64:(2 * t) % 64 + ht, (3 * t) % 64:(3 * t) % 64 + wd].contiguous().to(dev)
            torch.manual_seed(900 + t)
            n0 = slam.n
            slam(float(t), img, intr)
            slam.flush()
            if slam.is_initialized and t >= 8:
                dec, (s0, c0, s1, c1) = slam.last_keyframe
                assert bool(dec) == (slam.n == n0), t
                log.append((dec, 0.5 * (s0 / c0 + s1 / c1) if c0 > 0 and c1 > 0 else float("nan")))
        return slam, log
    finally:
        dpvo_mod._FRAME_CALL = fc_before


def test_unforced_keyframe_decision_on_the_device(dev):
    """the data-driven branch of track.hip:kf_decide (forced < 0: flow sums -> means -> threshold, dpvo.py:266-272) with a
    threshold INSIDE the range of the measured flows, so that it both drops and keeps -- against the Python-paced path that takes
    the same decision on the host from the same flow kernel.  Same decisions, same state, bit for bit; and the decision equals the
    reference's rule applied to the recorded flow."""
    _, probe = _run_unforced(dev, -1.0, frame_call=False)
    flows = np.array([f for _, f in probe])
    assert np.isfinite(flows).all() and not any(d for d, _ in probe)
    thr = float(np.median(flows))
    a, la = _run_unforced(dev, thr, frame_call=True, defer_keyframe=True, overlap_encoders=True)
    b, lb = _run_unforced(dev, thr, frame_call=False)
    torch.cuda.synchronize()
    da, db = [d for d, _ in la], [d for d, _ in lb]
    print(f"unforced keyframe test: threshold {thr:.4f}, device path dropped {sum(da)}/{len(da)}, host path {sum(db)}/{len(db)}")
    assert 3 <= sum(da) <= len(da) - 3, "the threshold must exercise both branches"
    assert da == db
    for (d, f) in la:
        assert bool(d) == (f < thr), (d, f, thr)                              # dpvo.py:272: m / 2 < KEYFRAME_THRESH
    assert a.n == b.n and a.m == b.m
    for k in ("ii", "jj", "kk", "ii_inac", "jj_inac", "kk_inac", "net", "target", "weight"):
        assert torch.equal(getattr(a.pg, k), getattr(b.pg, k)), k
    assert torch.equal(a.pg.poses_[:a.n], b.pg.poses_[:b.n]) and torch.equal(a.pg.patches_[:a.n], b.pg.patches_[:b.n])


def test_side_stream_draws_are_ordered_before_their_reader(dev):
    """with overlapped encoders the frame's three random draws (net.py:132-133, dpvo.py:427) are PRODUCED on the
    side stream, and the one-call frame path joins the encoders only in front of frame-state part 2 -- part 1 (coordinate / depth
    patches) reads the draws earlier.  With image_ready=None the side stream waits for the caller's stream position, so the draws and
    part 1 become runnable at the same moment: a stall on the main stream in front of the call (torch.cuda._sleep) makes that the
    normal case.  The patch coordinates the frame stored must be the draws of this frame's seed."""
    slam, _ = _run_unforced(dev, -1.0, n_frames=20, overlap_encoders=True, defer_keyframe=True)
    assert slam._fu is not None, "the one-call frame path must be the one exercised"
    ht, wd, seed = 96, 128, 11
    g = torch.Generator().manual_seed(seed)
    tex = torch.rand(3, ht + 64, wd + 64, generator=g)
    tex = torch.nn.functional.avg_pool2d(tex[None], 5, 1, 2)[0]
    tex = (255 * (tex - tex.min()) / (tex.max() - tex.min())).to(torch.uint8)
    intr = torch.tensor([100.0, 100.0, wd / 2, ht / 2], device=dev)
    for t in range(20, 32):
        img = tex[:, (2 * t) % 64:(2 * t) % 64 + ht, (3 * t) % 64:(3 * t) % 64 + wd].contiguous().to(dev)
        torch.cuda.synchronize()
        torch.cuda._sleep(30_000_000)                       # ~15-20 ms in front of everything this call enqueues
        torch.manual_seed(900 + t)
        slam(float(t), img, intr)                           # image_ready=None: the side stream waits behind the stall
        slam.flush()
        torch.manual_seed(900 + t)
        xs = torch.randint(1, wd // 4 - 1, size=[1, slam.M], device=dev)
        ys = torch.randint(1, ht // 4 - 1, size=[1, slam.M], device=dev)
        p = slam.pg.patches_[slam.n - 1]
        assert torch.equal(p[:, 0, 1, 1], xs[0].float()) and torch.equal(p[:, 1, 1, 1], ys[0].float()), t


def test_keyframe_step_decision_unit(dev):
    """dpvo_keyframe_step's decision from synthetic flow sums, incl. the empty-direction case: mean of an empty tensor is NaN in the
    reference (dpvo.py:264) and `NaN / 2 < thresh` is False -> keep"""
    a, _ = _run_unforced(dev, 1.0, n_frames=16, frame_call=True)
    fu = a._frame_update_buffers()
    import ctypes
    from dpvo_amd import _lib as L
    for (s0, c0, s1, c1), expect in (((4.0, 8.0, 4.0, 8.0), 1),       # means 0.5 + 0.5 -> m/2 = 0.5 < 1 -> drop
                                     ((40.0, 8.0, 4.0, 8.0), 0),      # 5 + 0.5 -> 2.75 -> keep
                                     ((8.0, 8.0, 8.0, 8.0), 0),       # exactly the threshold: strict '<' -> keep
                                     ((0.0, 0.0, 1.0, 8.0), 0),       # no edge i -> j: NaN -> keep
                                     ((1.0, 8.0, 0.0, 0.0), 0)):
        es = a.pg.edges
        par = 0 if es.a is fu["sets"][0] else 1
        args = fu["args"][par]
        kf = L.KeyframeStep.from_buffer_copy(args.kf)
        flow = torch.tensor([s0, c0, s1, c1, 0, 0, 0, 0] + [0.0] * 8, dtype=torch.float32, device=dev)
        res = torch.zeros(16 + 4 + 4 * (es.cap // 1024 + 2), dtype=torch.float32, device=dev)
        res[:16] = flow
        kf.flow4, kf.result, kf.result_host, kf.host_words = res.data_ptr(), res.data_ptr() + 32, None, 8
        kf.poses = a.pg.poses_.data_ptr()
        kf.E, kf.n, kf.forced, kf.n_ring = 0, a.n, -1, 0               # decision only: no edges to move, no rings to shift
        kf.inac_room = 0
        L.check(L.lib().dpvo_keyframe_step(ctypes.byref(kf), L.stream()), "dpvo_keyframe_step")
        torch.cuda.synchronize()
        assert int(res.view(torch.int32)[8].item()) == expect, ((s0, c0, s1, c1), expect)


def _run_loop_closure(dev, frame_call, n_frames=60, M=16, ht=96, wd=128, seed=13, **kw):
    """LOOP_CLOSURE=True (BASELINE config 5) on a stream that revisits its start, so that PatchGraph.edges_loop finds edges and
    update() runs global BAs; no scripted decision except the initialisation probe"""
    import dpvo_amd.dpvo as dpvo_mod
    fc_before = dpvo_mod._FRAME_CALL
    dpvo_mod._FRAME_CALL = frame_call
    try:
        cfg = base_cfg.clone(); cfg.merge_from_dict(DEFAULT_YAML)
        cfg.PATCHES_PER_FRAME, cfg.BUFFER_SIZE, cfg.KEYFRAME_THRESH, cfg.LOOP_CLOSURE = M, 256, -1.0, True
        torch.manual_seed(seed)
        slam = DPVO(cfg, VONet(), ht=ht, wd=wd, device=dev, **kw)
        slam.motion_probe = lambda: 1e9
        g = torch.Generator().manual_seed(seed)
        tex = torch.rand(3, ht + 64, wd + 64, generator=g)
        tex = torch.nn.functional.avg_pool2d(tex[None], 5, 1, 2)[0]
        tex = (255 * (tex - tex.min()) / (tex.max() - tex.min())).to(torch.uint8)
        intr = torch.tensor([100.0, 100.0, wd / 2, ht / 2], device=dev)
        fast = slow = 0
        for t in range(n_frames):
            img = tex[:, (2 * t) % 64:(2 * t) % 64 + ht, (3 * t) % 64:(3 * t) % 64 + wd].contiguous().to(dev)
            torch.manual_seed(700 + t)
            pend = slam._fu_pending
            slam(float(t), img, intr)
            fast += int(slam._fu_pending is not None and slam._fu_pending is not pend)
            slam.flush()
        return slam, fast
    finally:
        dpvo_mod._FRAME_CALL = fc_before


def test_loop_closure_on_the_one_call_path(dev):
    """BASELINE config 5: with LOOP_CLOSURE=True the one-call frame path serves every frame that takes update()'s local-BA branch
    (no long-range edge active, none appended), the call-by-call path the others (loop edges appended in front of the frame's own,
    global BA over active + inactive edges).  Against the all-call-by-call run: the same state bit for bit -- which also needs the
    global BA to be bit-repeatable (gba_row_kernel, round 4)."""
    a, fast = _run_loop_closure(dev, True, defer_keyframe=True, overlap_encoders=True)
    b, none = _run_loop_closure(dev, False)
    torch.cuda.synchronize()
    gb = int(a.ran_global_ba.sum())
    print(f"loop closure: {fast} of 60 frames on the one-call path, {gb} global BA runs, {a.pg.ii_inac.numel()} inactive edges")
    assert none == 0 and fast >= 15, "both paths must have been exercised"
    # the one-call path runs edges_loop's candidate test in the TAIL of the previous frame's call (dpvo_frame_update_t.loop_out): same
    # kernel, same state, same ranges -- the states below are bit-identical to the run that launches it in front of every frame
    print(f"   candidate tests served by the previous call's tail: {a.pg.loop_pre_hits} (call-by-call run: {b.pg.loop_pre_hits})")
    assert a.pg.loop_pre_hits >= 5 and b.pg.loop_pre_hits == 0
    assert gb >= 2 and gb == int(b.ran_global_ba.sum()) and np.array_equal(a.ran_global_ba, b.ran_global_ba)
    assert a.n == b.n and a.m == b.m and a.last_global_ba == b.last_global_ba
    for k in ("ii", "jj", "kk", "ii_inac", "jj_inac", "kk_inac", "net", "target", "weight", "target_inac", "weight_inac"):
        assert torch.equal(getattr(a.pg, k), getattr(b.pg, k)), k
    assert torch.equal(a.pg.poses_[:a.n], b.pg.poses_[:b.n]) and torch.equal(a.pg.patches_[:a.n], b.pg.patches_[:b.n])
    pa, _ = a.terminate(); pb, _ = b.terminate()
    assert np.array_equal(pa, pb)


def test_edge_store_deferred_compaction(dev):
    """EdgeStore.keep(defer_net=True): every reader of `net` other than the fused update operator sees compact rows"""
    from dpvo_amd.patchgraph import EdgeStore
    g = torch.Generator().manual_seed(2)
    def fill(es, n):
        ii = torch.randint(0, 30, (n,), generator=g).to(dev); jj = torch.randint(0, 30, (n,), generator=g).to(dev)
        kk = torch.randint(0, 900, (n,), generator=g).to(dev)
        es.append(ii, jj, kk, net=torch.randn(n, es.D, generator=g).to(dev), target=torch.randn(n, 2, generator=g).to(dev),
                  weight=torch.rand(n, 2, generator=g).to(dev))
    a, b = EdgeStore(384, dev, cap=4096), EdgeStore(384, dev, cap=4096)
    for es in (a, b):
        g.manual_seed(2); fill(es, 1000)
    keep1 = torch.sort(torch.randperm(1000, generator=g)[:700]).values.to(dev)
    a.keep(keep1, defer_net=True); b.keep(keep1)
    assert a.net_pending is not None and a.net_pending[1] == 700
    buf, rows, n_kept, whole = a.net_deferred()
    assert buf.data_ptr() == whole.data_ptr() and buf.shape[0] == 700
    assert torch.equal(whole[rows[:n_kept]], b.view("net"))                       # what the update operator's first kernel reads
    keep2 = torch.sort(torch.randperm(700, generator=g)[:500]).values.to(dev)
    a.keep(keep2, defer_net=True); b.keep(keep2)                                 # second removal: the first one is applied now
    assert a.net_pending[1] == 500
    for es in (a, b):
        g.manual_seed(7); fill(es, 64)                                           # an append with explicit state: applied first
    assert a.net_pending is None and a.E == b.E == 564
    for name in ("ii", "jj", "kk", "net", "target", "weight"):
        assert torch.equal(a.view(name), b.view(name)), name
    a.keep(keep2, defer_net=True); b.keep(keep2)
    assert torch.equal(a.view("net"), b.view("net")) and a.net_pending is None    # view() materialises


def test_run_to_run_determinism(dev):
    decisions = [(True, False)] * 14
    a, _, _ = _run(dev, decisions, seed=3)
    b, _, _ = _run(dev, decisions, seed=3)
    assert torch.equal(a.pg.poses_[:a.n], b.pg.poses_[:b.n])
    assert torch.equal(a.pg.patches_[:a.n], b.pg.patches_[:b.n])
    assert torch.equal(a.pg.net, b.pg.net)


def test_loop_closure_global_ba_path(dev):
    """BASELINE config 5 plumbing (LOOP_CLOSURE=True): edges_loop -> reduce_edges -> append, normalize(), global BA."""
    from oracle.graph_ref import GraphRef  # noqa: F401  (import check only)
    cfg = base_cfg.clone(); cfg.merge_from_dict(DEFAULT_YAML)
    cfg.PATCHES_PER_FRAME = 16
    cfg.BUFFER_SIZE = 256
    cfg.LOOP_CLOSURE = True
    cfg.MAX_EDGE_AGE = 100
    cfg.KEYFRAME_THRESH = -1.0
    torch.manual_seed(0)
    ht, wd = 96, 128
    slam = DPVO(cfg, VONet(), ht=ht, wd=wd, device=dev)
    slam.motion_probe = lambda: 1e9
    g = torch.Generator().manual_seed(0)
    intr = torch.tensor([100.0, 100.0, wd / 2, ht / 2], device=dev)
    saw_loop = False
    for t in range(75):
        img = torch.randint(0, 255, (3, ht, wd), generator=g, dtype=torch.uint8).to(dev)
        slam(float(t), img, intr)
        if slam.pg.ii.numel():
            saw_loop = saw_loop or bool(((slam.pg.jj - slam.pg.ii) > 30).any())
    assert saw_loop, "no loop-closure edges were created"
    assert slam.ran_global_ba.any(), "global BA never ran"
    n = slam.n
    assert torch.isfinite(slam.pg.poses_[:n]).all() and torch.isfinite(slam.pg.patches_[:n]).all()
    assert (slam.pg.patches_[:n, :, 2] > 0).all()
    poses, tstamps = slam.terminate()
    assert poses.shape == (75, 7) and np.isfinite(poses).all()


def _graph_golden():
    import os
    return np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "graph.npz"))


def test_bookkeeping_matches_the_reference_dpvo_class(dev):
    """The tracker's integer state against the state the REFERENCE'S OWN DPVO class went through (tests/golden/graph.npz, made by
    tests/golden/make_golden_graph.py from the imported dpvo/dpvo.py with the float pipeline replaced by the same scripted
    decisions): frame / patch counters, active and inactive edge lists, timestamps after every frame, removed-frame links."""
    from oracle.graph_ref import GraphRef  # noqa: F401
    g = _graph_golden()
    decisions = [(bool(a), bool(d)) for a, d in g["dpvo_decisions"]]
    M = int(g["dpvo_M"])
    cfg = base_cfg.clone(); cfg.merge_from_dict(DEFAULT_YAML)
    cfg.PATCHES_PER_FRAME = M
    cfg.BUFFER_SIZE = 256
    torch.manual_seed(0)
    ht, wd = 96, 128
    slam = DPVO(cfg, VONet(), ht=ht, wd=wd, device=dev)
    gen = torch.Generator().manual_seed(0)
    intr = torch.tensor([100.0, 100.0, wd / 2, ht / 2], device=dev)
    state = {}
    slam.motion_probe = lambda: 1e9 if state["accept"] else 0.0
    orig = pops.motionmag_pair
    thresh = cfg.KEYFRAME_THRESH

    def fake(*a, defer=False, host_buf=None, **k):
        res = (0.0, 0.0) if state["drop"] else (4 * thresh, 4 * thresh)
        return (lambda: res) if defer else res
    pops.motionmag_pair = fake
    slam.keyframe_override = lambda counter: state["drop"]
    o = {"E": 0, "Ei": 0, "t": 0}
    try:
        for t, (accept, drop) in enumerate(decisions):
            state["accept"], state["drop"] = accept, drop
            img = torch.randint(0, 255, (3, ht, wd), generator=gen, dtype=torch.uint8).to(dev)
            slam(float(t), img, intr)
            slam.flush()
            E, Ei, n = int(g["dpvo_E"][t]), int(g["dpvo_E_inac"][t]), int(g["dpvo_n"][t])
            assert (slam.n, slam.m, slam.counter) == (n, int(g["dpvo_m"][t]), int(g["dpvo_counter"][t])), t
            for k in ("ii", "jj", "kk"):
                assert np.array_equal(getattr(slam.pg, k).cpu().numpy(), g["dpvo_" + k][o["E"]:o["E"] + E]), (t, k)
                assert np.array_equal(getattr(slam.pg, k + "_inac").cpu().numpy(),
                                      g["dpvo_" + k + "_inac"][o["Ei"]:o["Ei"] + Ei]), (t, k)
            assert np.array_equal(slam.pg.tstamps_[:n], g["dpvo_tstamps"][o["t"]:o["t"] + n]), t
            o["E"] += E; o["Ei"] += Ei; o["t"] += n
    finally:
        pops.motionmag_pair = orig
    assert sorted(int(k) for k in slam.pg.delta.keys()) == g["dpvo_delta_keys"].tolist()
    assert [int(slam.pg.delta[k][0]) for k in sorted(slam.pg.delta.keys())] == g["dpvo_delta_t0"].tolist()


def test_patchgraph_edges_loop_and_normalize_match_the_reference(dev):
    """PatchGraph.edges_loop (patchgraph.py:56-82: candidate generation, flow test, the reference's reduce_edges) and .normalize
    (:84-95) on the synthetic loop-closure state of tests/golden/make_golden_graph.py: the loop edges must be the reference's
    bit for bit, the rescaled poses / depths / points agree to f32 rounding (the reference side ran in f32 on f64 lietorch stubs)."""
    from types import SimpleNamespace
    from dpvo_amd.patchgraph import PatchGraph
    from dpvo_amd.lietorch import SE3
    g = _graph_golden()
    M, n = int(g["pg_M"]), int(g["pg_n"])
    cfg = SimpleNamespace(PATCHES_PER_FRAME=M, BUFFER_SIZE=64, LOOP_CLOSURE=True, REMOVAL_WINDOW=22, GLOBAL_OPT_FREQ=15,
                          KEYFRAME_INDEX=4, MAX_EDGE_AGE=1000, BACKEND_THRESH=64.0)
    pg = PatchGraph(cfg, 3, 384, 1000, device=dev, dtype=torch.float)
    # the active edge store holds the window's edges AND the largest batch edges_loop can return (1000 frame pairs x M) without growing:
    # a doubling inside a tracked frame cost one frame of 9.9 ms (profiles/r06_h_lc_host_trace.txt)
    from dpvo_amd.patchgraph import MAX_LOOP_PAIRS
    assert pg.edges.cap >= M * 24 * 28 + MAX_LOOP_PAIRS * M
    cfg0 = SimpleNamespace(**{**vars(cfg), "LOOP_CLOSURE": False})
    assert PatchGraph(cfg0, 3, 384, 1000, device=dev, dtype=torch.float).edges.cap == max(1 << 16, 1 << (M * 24 * 28 - 1).bit_length())
    pg.n, pg.m = n, n * M
    pg.poses_[:n] = torch.from_numpy(g["pg_poses"]).to(dev)
    pg.patches_[:n] = torch.from_numpy(g["pg_patches"]).to(dev).view(n, M, 3, 3, 3)
    pg.intrinsics_[:n] = torch.from_numpy(g["pg_intr"]).to(dev)
    for f in range(64):
        pg.index_[f] = f
    kk, jj = pg.edges_loop()
    assert np.array_equal(kk.cpu().numpy(), g["pg_loop_kk"]) and np.array_equal(jj.cpu().numpy(), g["pg_loop_jj"])
    assert g["pg_loop_kk"].size > 0
    pg.delta[7] = (6, SE3(torch.tensor([[0.1, -0.2, 0.05, 0.0, 0.0, 0.0, 1.0]], device=dev)))
    pg.normalize()
    from tests import helpers as H
    H.assert_close(pg.poses_[:n].cpu().numpy(), g["pg_norm_poses"], 2e-5, 2e-5, "normalize: poses")
    H.assert_close(pg.patches_[:n].cpu().numpy(), g["pg_norm_patches"], 2e-5, 2e-5, "normalize: patches")
    H.assert_close(pg.points_[:n * M].cpu().numpy(), g["pg_norm_points"], 2e-4, 2e-4, "normalize: points")
    H.assert_close(pg.delta[7][1].data.cpu().numpy(), g["pg_norm_delta"], 2e-6, 2e-6, "normalize: delta")


def test_loop_flow_kernel_matches_the_reference_composition(dev):
    """dpvo_loop_flow (one launch, PatchGraph.edges_loop's candidate test) against the reference's own sequence of operations
    (patchgraph.py:56-72: flatmeshgrid, pops.flow_mag on the centre pixels, validity mask, masked mean, the 0.75 M rule) built from
    dpvo_flow_mag + torch: same +inf pattern, values to f32 rounding of a 96-term sum (the reduction order is the only difference)."""
    import ctypes
    from dpvo_amd import _lib as L
    from dpvo_amd import projective_ops as pops
    from dpvo_amd import synthetic as S
    from dpvo_amd.utils import flatmeshgrid
    N, M, P = 60, 96, 3
    poses, patches, intr = S.make_scene(N, M, ht=120, wd=160, seed=5, noise=0.02)[:3]
    poses, patches, intr = poses.to(dev), patches.to(dev).view(N * M, 3, P, P).clone(), intr.to(dev)
    # invalid pixels (Z <= 0.2 after the transform): inverse depths of +-40 put a point behind the camera for one sign of the baseline's
    # z component -- sprinkled over all frames, and ALL of frame 5 (half of them invalid whichever sign: under the 0.75 M rule -> +inf)
    patches[::7, 2] = 40.0
    patches[3::7, 2] = -40.0
    patches[5 * M:6 * M:2, 2] = 40.0
    patches[5 * M + 1:6 * M:2, 2] = -40.0
    ix = torch.arange(N, device=dev)[:, None].expand(N, M).contiguous()
    j0, n_j, i0, n_i = 45, 11, 3, 30
    out = torch.full((n_j * n_i,), -1.0, device=dev)
    L.check(L.lib().dpvo_loop_flow(L.ptr(poses), L.ptr(patches), L.ptr(intr), L.ptr(ix), L.i64(j0), L.i64(n_j), L.i64(i0), L.i64(n_i),
                                   L.i32(M), L.i32(P), L.f32(0.5), L.ptr(out), L.stream()), "dpvo_loop_flow")
    jj, kk = flatmeshgrid(torch.arange(j0, j0 + n_j, device=dev), torch.arange(i0 * M, (i0 + n_i) * M, device=dev), indexing="ij")
    ii = ix.view(-1)[kk]
    centre = patches[..., P // 2, P // 2].reshape(1, -1, 3, 1, 1)
    flow_mg, nval = pops.flow_mag(poses[None], centre, intr[None], ii, jj, kk, beta=0.5)
    val = (nval > 0.5).float()
    s = (flow_mg * val).view(-1, M).sum(dim=1).float()
    c = val.view(-1, M).sum(dim=1).clamp(min=1)
    ref = torch.where(c > (M * 0.75), s / c, torch.full_like(c, float("inf")))
    a, b = out.cpu().numpy(), ref.cpu().numpy()
    assert np.array_equal(np.isinf(a), np.isinf(b)) and 0 < np.isinf(b).sum() < b.size
    fin = ~np.isinf(b)
    assert np.allclose(a[fin], b[fin], rtol=2e-6, atol=1e-6), np.abs(a[fin] - b[fin]).max()
    # ... and against the ORACLE (oracle/: the f64 C restatement of pops.flow_mag, then patchgraph.py:64-72 in numpy): the +inf pattern may
    # differ only where a point sits on the validity threshold in f32 vs f64 (none here), the values agree to f32 rounding
    import oracle
    of, ov = oracle.flow_mag(poses.cpu().numpy(), centre.cpu().numpy().reshape(-1, 3, 1, 1), intr.cpu().numpy(),
                             ii.cpu().numpy(), jj.cpu().numpy(), kk.cpu().numpy(), beta=0.5)
    of, ov = of.reshape(-1, M), ov.reshape(-1, M).astype(np.float64)
    oc = ov.sum(1)
    with np.errstate(invalid="ignore", divide="ignore"):
        oref = np.where(oc > 0.75 * M, (of * ov).sum(1) / np.maximum(oc, 1.0), np.inf)
    assert np.array_equal(np.isinf(a), np.isinf(oref))
    assert np.allclose(a[fin], oref[fin], rtol=2e-4, atol=1e-4), np.abs(a[fin] - oref[fin]).max()
    # ... and into pinned host memory, as edges_loop uses it
    host = torch.empty(n_j * n_i, dtype=torch.float32).pin_memory()
    L.check(L.lib().dpvo_loop_flow(L.ptr(poses), L.ptr(patches), L.ptr(intr), L.ptr(ix), L.i64(j0), L.i64(n_j), L.i64(i0), L.i64(n_i),
                                   L.i32(M), L.i32(P), L.f32(0.5), ctypes.c_void_p(host.data_ptr()), L.stream()), "dpvo_loop_flow")
    torch.cuda.synchronize()
    assert np.array_equal(host.numpy(), a)
    # edge cases of the entry: an empty candidate range is a no-op, nonsense is refused (never a launch with a zero / negative grid)
    args = lambda n_j_, n_i_, j0_=j0: (L.ptr(poses), L.ptr(patches), L.ptr(intr), L.ptr(ix), L.i64(j0_), L.i64(n_j_), L.i64(i0), L.i64(n_i_),
                                       L.i32(M), L.i32(P), L.f32(0.5), L.ptr(out), L.stream())
    before = out.clone()
    assert L.lib().dpvo_loop_flow(*args(0, n_i)) == 0 and L.lib().dpvo_loop_flow(*args(n_j, 0)) == 0
    torch.cuda.synchronize()
    assert torch.equal(out, before)
    assert L.lib().dpvo_loop_flow(*args(-1, n_i)) != 0 and L.lib().dpvo_loop_flow(*args(n_j, n_i, -3)) != 0
    assert L.lib().dpvo_loop_flow(L.ptr(None), L.ptr(patches), L.ptr(intr), L.ptr(ix), L.i64(j0), L.i64(n_j), L.i64(i0), L.i64(n_i),
                                  L.i32(M), L.i32(P), L.f32(0.5), L.ptr(out), L.stream()) != 0


def test_tracker_does_not_load_the_comparator_library(dev):
    """the product path is libdpvo_hip.so alone: a tracker run (initialisation, steady-state frames through the one-call frame
    path, terminate) never maps libdpvo_hip_cmp.so (the launch-by-launch / patch-major update operators are test partners)"""
    import subprocess, sys, os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = """
import torch
from dpvo_amd.config import cfg as base_cfg, DEFAULT_YAML
from dpvo_amd.dpvo import DPVO
from dpvo_amd.net import VONet
cfg = base_cfg.clone(); cfg.merge_from_dict(DEFAULT_YAML); cfg.PATCHES_PER_FRAME = 16; cfg.BUFFER_SIZE = 128
torch.manual_seed(0)
slam = DPVO(cfg, VONet(), ht=96, wd=128, device=torch.device("cuda:0"), defer_keyframe=True, overlap_encoders=True)
slam.motion_probe = lambda: 1e9
g = torch.Generator().manual_seed(0)
intr = torch.tensor([100.0, 100.0, 64.0, 48.0], device="cuda:0")
for t in range(24):
    slam(float(t), torch.randint(0, 255, (3, 96, 128), generator=g, dtype=torch.uint8).cuda(), intr)
poses, _ = slam.terminate()
maps = open("/proc/self/maps").read()
assert "libdpvo_hip.so" in maps and "libdpvo_hip_cmp" not in maps, "comparator library mapped"
print("ok", poses.shape)
"""
    out = subprocess.run([sys.executable, "-c", code], cwd=root, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "ok" in out.stdout, out.stderr[-2000:]


def test_plan_on_the_side_stream_is_bit_identical(dev):
    """dpvo_frame_update_t.plan_stream (DPVO_PLAN_ASIDE=1, off by default): the graph plan issued on the encoders' stream, forked behind
    the new frame's edges and joined in front of the update operator -- same tracker state, bit for bit, with unscripted keyframe
    decisions (the flow test reads the plan's edge list) and with a third stream"""
    import dpvo_amd.dpvo as dpvo_mod
    _, probe = _run_unforced(dev, -1.0, n_frames=30, overlap_encoders=True, defer_keyframe=True)
    thr = float(np.median([f for _, f in probe]))
    before = (dpvo_mod._PLAN_ASIDE, dpvo_mod._PLAN_OWN_STREAM)
    runs = []
    try:
        for aside, own in ((False, False), (True, False), (True, True)):
            dpvo_mod._PLAN_ASIDE, dpvo_mod._PLAN_OWN_STREAM = aside, own
            slam, log = _run_unforced(dev, thr, n_frames=40, overlap_encoders=True, defer_keyframe=True)
            torch.cuda.synchronize()
            runs.append((slam, log))
            if aside:
                assert slam._fu is not None and slam._fu.get("ev_plan") is not None, "the side-stream plan was not exercised"
    finally:
        dpvo_mod._PLAN_ASIDE, dpvo_mod._PLAN_OWN_STREAM = before
    a = runs[0][0]
    assert any(d for d, _ in runs[0][1]) and not all(d for d, _ in runs[0][1])
    for b, log in runs[1:]:
        assert [d for d, _ in log] == [d for d, _ in runs[0][1]] and a.n == b.n
        for k in ("ii", "jj", "kk", "net", "target", "weight"):
            assert torch.equal(getattr(a.pg, k), getattr(b.pg, k)), k
        assert torch.equal(a.pg.poses_[:a.n], b.pg.poses_[:b.n]) and torch.equal(a.pg.patches_[:a.n], b.pg.patches_[:b.n])

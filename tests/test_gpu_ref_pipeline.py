"""Tracker-level parity against the REFERENCE'S OWN tracker on the MI355X, at the configuration bench.py times.

The other side of every comparison here is `dpvo.dpvo.DPVO` itself (dpvo/dpvo.py:20-473) with its own net.py / patchgraph.py /
projective_ops.py / blocks.py / lietorch Python and its own native kernels (cuda_corr, cuda_ba compiled for gfx950 by
oracle/build_ref.py), run through oracle/ref_pipeline.py with torch stand-ins for torch_scatter / lietorch_backends
(oracle/ref_standins.py, pinned in tests/test_oracle.py).  Both trackers get the same random-init weights (strict state-dict load),
the same 480x640 frames, and the same random draws (same torch seed before each call: patch centroids net.py:132-133, depth
initialisation dpvo.py:427).  Our side runs exactly what bench.py switches on: one C-ABI call per frame (dpvo_frame_update), the
next frame's encoders on a second stream held behind the update operator, the keyframe record resolved one call later.

What can be asserted, and why it is split the way it is (measured: profiles/r04_ref_parity_pipeline.txt):
  * the INTEGER state (frame / patch counters, edge lists, inactive lists, timestamps, patch coordinates, colours) does not depend on
    float noise while no keyframe test is near its threshold: asserted bit-exact on every frame, free running, >= 70 frames;
  * the FLOAT state of a tracker with random weights is a chaotic recurrence: a static camera makes depth / scale unobservable, the
    flow head emits noise, and around frame 30 the trajectory runs away (extent 0.02 -> 40).  The reference run TWICE on identical
    inputs drifts apart just as fast (float atomics in ba_cuda.cu:335-373; tools/ref_parity.py scenario R: 4e-6 until frame 30,
    6e-2 .. 11 afterwards), so an accumulated distance after the run-away says nothing about an implementation.  Hence:
      - free running, the poses must agree to 1e-3 (north_star's figure) up to the run-away (frames < 28; measured <= 1e-4);
      - with teacher forcing (after each frame our float state is reset to the reference's), EVERY frame of the whole run is a
        one-step comparison at E = 45 312: poses to 1e-3 with at most two outlier frames (the run-away frame itself, where the
        reference's own BA step is singular: a 1e-2 px target difference moves a pose by 5e-2 there), flows to 1e-3 px.
The tolerances are written where they are asserted; the measured values are printed (-s) and committed under profiles/."""
import numpy as np
import pytest
import torch

from tests import ref_harness as H

pytestmark = pytest.mark.gpu

HT, WD, M = 480, 640, 96        # BASELINE config 2: what bench.py times
POSE_TOL = 1e-3                 # north_star: "ATE within 1e-3 m of reference"; applied to every pose component, every frame
FLOW_TOL = 1e-3                 # px, keyframe flow test input (dpvo.py:257-270) under teacher forcing (measured 4e-5)


@pytest.fixture(scope="module")
def RP():
    from oracle import ref_pipeline
    if not ref_pipeline.available():
        pytest.skip("oracle/_ref not built (oracle/build_ref.py needs /root/reference)")
    return ref_pipeline


@pytest.fixture(scope="module")
def stream(dev):
    frames = H.stream(64, HT, WD, dev)
    intr = torch.tensor([320.0, 320.0, 320.0, 240.0], device=dev)
    return frames, intr


def _int_exact(recs, n_frames):
    s = H.summarise(recs)
    assert s["frames"] == n_frames and s["int_equal_frames"] == n_frames, s["first_int_mismatch"]
    assert s["patch_xy_equal"] and s["intrinsics_equal"] and s["colors_maxdiff"] == 0 and s["finite"]
    return s


def _pose_budget(recs, tol=POSE_TOL, max_outliers=2):
    """every frame: max |pose component difference| <= tol x max(1, extent of the trajectory so far) -- 1e-3 absolute while the
    trajectory is metre-sized, 1e-3 relative once the random-weight tracker has run away to tens of units; at most `max_outliers`
    frames may exceed it (measured: one, the run-away frame t = 32, where the reference's own BA step is singular)"""
    out = [(r["t"], r["pose_max"], r["extent"]) for r in recs if r.get("pose_max", 0.0) > tol * max(1.0, r.get("extent", 0.0))]
    assert len(out) <= max_outliers, f"frames with a pose component off by more than {tol} x max(1, extent): {out}"
    return out


def test_free_running_bench_configuration(dev, RP, stream):
    """70 frames, E = 45 312 from frame 44 on, no keyframe dropped (bench.py's workload), both trackers free running"""
    frames, intr = stream
    n_frames = 70
    ours, theirs, _ = H.build_pair(dev, HT, WD, M, KEYFRAME_THRESH=-1.0)
    recs = H.run_lockstep(ours, theirs, frames, n_frames, intr, feed=True)
    s = _int_exact(recs, n_frames)
    assert s["E_last"] == 45312 and ours._fu is not None, "the one-call frame path must have been the one that ran"
    print(f"\nfree running: integer state bit-exact on {s['int_equal_frames']}/{n_frames} frames; max pose distance before the run-away "
          f"(t < 28) {s['pose_max_first28']:.3e}, over the whole run {s['pose_max']:.3e} on a trajectory of extent {s['extent_last']:.3g}; "
          f"flow test inputs differ by <= {s['flow_absdiff_max']:.3e} px; series (t, distance, extent): {s['pose_series']}")
    # (t < 24: the chaotic amplification sets in between frames 20 and 30 and its onset moves with the reference's own float-atomics
    #  noise from run to run -- measured over five runs: 4e-5 .. 2.8e-4 up to t = 28, never above 1e-4 up to t = 24)
    assert s["pose_max_first24"] < POSE_TOL
    # ... and what bench.py's loop does (no flush between frames: every record resolved one call later) ends in the same bits
    final = RP.snapshot(ours)
    del theirs
    b, unused, _ = H.build_pair(dev, HT, WD, M, KEYFRAME_THRESH=-1.0)
    del unused
    with torch.no_grad():
        for t in range(n_frames):
            torch.manual_seed(5000 + t)
            b(float(t), frames[t % frames.shape[0]], intr, image_ready=False)
        b.flush()
    sb = RP.snapshot(b)
    for k in ("ii", "jj", "kk", "poses", "patches"):
        assert np.array_equal(sb[k], final[k]), k
    assert torch.equal(b.pg.net, ours.pg.net)


def test_teacher_forced_bench_configuration(dev, RP, stream):
    """every frame of an 80-frame run at the bench configuration as a one-step comparison (see module docstring)"""
    frames, intr = stream
    ours, theirs, _ = H.build_pair(dev, HT, WD, M, KEYFRAME_THRESH=-1.0)
    recs = H.run_lockstep(ours, theirs, frames, 80, intr, feed=True, teacher=True)
    s = _int_exact(recs, 80)
    out = _pose_budget(recs)
    print(f"\nteacher forced: |pose| <= {POSE_TOL} on {80 - len(out)}/80 frames, outliers {out}; typical (median over frames) "
          f"{np.median([r['pose_max'] for r in recs if 'pose_max' in r]):.2e}; hidden state max {s['net_max']:.2e} rms {s['net_rms']:.2e}; "
          f"target {s['target_max']:.2e} px, weight {s['weight_max']:.2e}; flow {s['flow_absdiff_max']:.2e} px; "
          f"depth rel. p50 {s['depth_rel_p50']:.2e} p90 {s['depth_rel_p90']:.2e}")
    assert s["E_last"] == 45312
    assert s["flow_absdiff_max"] < FLOW_TOL
    # the update operator's outputs against the reference's (f16 GEMMs on both sides, f16 correlation accumulate on the reference's):
    # hidden state 2e-2 (f16 ulp at |net| ~ 8), BA targets 2e-2 px, confidence weights 2e-3 -- measured 6e-3 / 8e-3 / 7e-4
    assert s["net_max"] < 2e-2 and s["net_rms"] < 2e-3 and s["target_max"] < 2e-2 and s["weight_max"] < 2e-3
    assert s["depth_rel_p50"] < 5e-3


def test_teacher_forced_end_to_end_encoders(dev, RP, stream):
    """as above, but the reference also runs its OWN encoders (torch / MIOpen convolutions under autocast) instead of being fed ours:
    the stated difference is the encoders' f16 arithmetic (tests/test_gpu_encoders.py: a few f16 ulps per feature)"""
    frames, intr = stream
    ours, theirs, _ = H.build_pair(dev, HT, WD, M, feed=False, KEYFRAME_THRESH=-1.0)
    recs = H.run_lockstep(ours, theirs, frames, 60, intr, feed=False, teacher=True)
    s = _int_exact(recs, 60)
    out = _pose_budget(recs)
    print(f"\nend to end: |pose| <= {POSE_TOL} on {60 - len(out)}/60 frames, outliers {out}; hidden state max {s['net_max']:.2e} rms "
          f"{s['net_rms']:.2e}; target {s['target_max']:.2e} px; flow {s['flow_absdiff_max']:.2e} px")
    # (flow magnitudes are 0.2 .. 8 px here; with different encoder arithmetic on the two sides they agree to 2e-2 px: measured 3e-3)
    assert s["flow_absdiff_max"] < 2e-2
    assert s["net_max"] < 4e-2 and s["net_rms"] < 4e-3 and s["target_max"] < 4e-2


def test_unscripted_keyframe_decisions(dev, RP, stream):
    """KEYFRAME_THRESH inside the range of the flow magnitudes this stream produces, NO override on either side: the device-side
    decision of track.hip:kf_decide (flow sums -> mean -> threshold, dpvo.py:266-272) against the reference's Python, 70 frames,
    teacher forced so that the run goes on past a hypothetical knife-edge frame with both trackers in the same state.
    0.58 px = the median flow of the no-drop run (tools/ref_parity.py scenario A)."""
    frames, intr = stream
    thr = 0.58
    ours, theirs, _ = H.build_pair(dev, HT, WD, M, KEYFRAME_THRESH=thr)
    assert ours.keyframe_override is None
    recs = H.run_lockstep(ours, theirs, frames, 70, intr, feed=True, teacher=True)
    s = H.summarise(recs)
    dec = [(r["t"], r["drop_ours"], r["drop_ref"], r["flow_ours"], r["flow_ref"]) for r in recs if r.get("flow_ref") is not None]
    drops = sum(1 for d in dec if d[2])
    margin = min(abs(d[4] - thr) for d in dec)
    print(f"\nunscripted decisions: {len(dec)} decisions, {drops} keyframes dropped by the reference, all agree: "
          f"{all(d[1] == d[2] for d in dec)}; smallest |flow - threshold| {margin:.2e} px, largest |flow_ours - flow_ref| "
          f"{s['flow_absdiff_max']:.2e} px")
    # A decision may differ only on a knife edge: the reference's flow within 1e-3 px of the threshold (its own flows move by more than
    # that between two of its runs; ours differ from them by 3-6e-5 px).  The run stops at such a frame -- the integer states part
    # there by definition -- and everything before it must be exact.  (Four runs so far: no such frame, smallest margin 1.2e-3 px.)
    bad = s["first_decision_mismatch"]
    assert bad is None or abs(bad["flow_ref"] - thr) < 1e-3, bad
    ok_frames = len(recs) if bad is None else len(recs) - 1
    assert s["int_equal_frames"] >= ok_frames and (bad is not None or s["int_equal_frames"] == 70), s["first_int_mismatch"]
    assert len(dec) >= 40 and drops >= 10 and len(dec) - drops >= 10, "both branches of dpvo.py:272 must be exercised"
    assert s["flow_absdiff_max"] < FLOW_TOL
    _pose_budget(recs)


def test_loop_closure_configuration(dev, RP, stream):
    """BASELINE config 5 (LOOP_CLOSURE=True): loop edges from PatchGraph.edges_loop (thresholded + NMS'd flow magnitudes), edges kept
    alive by the lc rule of dpvo.py:307-308, global BA over active + inactive edges (dpvo.py:312-326, EfficentE on the reference's
    side) -- 85 frames, teacher forced"""
    frames, intr = stream
    ours, theirs, _ = H.build_pair(dev, HT, WD, M, buffer=512, LOOP_CLOSURE=True, KEYFRAME_THRESH=-1.0)
    recs = H.run_lockstep(ours, theirs, frames, 85, intr, feed=True, teacher=True)
    s = _int_exact(recs, 85)
    out = _pose_budget(recs)
    gb_o, gb_r = int(ours.ran_global_ba.sum()), int(theirs.ran_global_ba.sum())
    print(f"\nloop closure: integer state (incl. loop edges) bit-exact on 85/85 frames, {gb_r} global BA runs on each side, "
          f"{int(theirs.pg.ii_inac.numel())} inactive edges; |pose| <= {POSE_TOL} on {85 - len(out)}/85 frames, outliers {out}")
    assert gb_o == gb_r >= 2
    assert int((theirs.pg.jj - theirs.pg.ii > 30).sum()) > 0 or int(theirs.pg.ii_inac.numel()) > 0

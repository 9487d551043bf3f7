"""Tracker-level parity against the REFERENCE'S OWN tracker on the MI355X, at the configuration bench.py times.

(Named test_zz_*: the reference's bundle adjustment is not deterministic -- float atomics, ba_cuda.cu:335-373 -- so this file
collects LAST: under `pytest -x` nothing else can be hidden behind it.  VERDICT r4 1d.)

The other side of every comparison here is `dpvo.dpvo.DPVO` itself (dpvo/dpvo.py:20-473) with its own net.py / patchgraph.py /
projective_ops.py / blocks.py / lietorch Python and its own native kernels (cuda_corr, cuda_ba compiled for gfx950 by
oracle/build_ref.py), run through oracle/ref_pipeline.py with torch stand-ins for torch_scatter / lietorch_backends
(oracle/ref_standins.py, pinned in tests/test_oracle.py).  Both trackers get the same random-init weights (strict state-dict load),
the same 480x640 frames, and the same random draws (same torch seed before each call: patch centroids net.py:132-133, depth
initialisation dpvo.py:427).  Our side runs exactly what bench.py switches on: one C-ABI call per frame (dpvo_frame_update), the
next frame's encoders on a second stream held behind the update operator, the keyframe record resolved one call later.

What is asserted, and why a failure can only have an implementation reason:
  * the INTEGER state (frame / patch counters, edge lists, inactive lists, timestamps, patch coordinates, colours): bit-exact on
    every frame of every scenario;
  * the UPDATE OPERATOR's outputs under teacher forcing (after each frame our float state is reset to the reference's, so every
    frame is a one-step comparison at E = 45 312): hidden state, BA targets, confidence weights and the keyframe flow test's input
    within stated f16-level tolerances on EVERY frame -- these do not depend on the conditioning of anything;
  * the POSES: a tracker with random weights on a static scene is a chaotic recurrence (depth / scale unobservable, the flow head
    emits noise; around frame 30 the trajectory runs away and the reference's own Gauss-Newton step becomes singular: a 1e-2 px
    target difference moves a pose by 5e-2 there, and the reference run twice drifts as far apart: tools/ref_parity.py scenario R).
    A constant "at most two bad frames" budget (round 4) cannot tell that from a bug.  Now, per frame (tests/ref_harness.py:attribute):
      - yard      = the reference's bundle adjustment re-run on ITS OWN captured inputs, 3 x, against its own result: the
                    reference-vs-reference one-step spread, measured in this test on this box;
      - ba_dist   = OUR bundle adjustment on the reference's captured inputs against the reference's result
                    -> asserted <= max(1e-3 x max(1, extent), K x yard) on every frame: the BA is attributed by itself;
      - pose_max  = our tracker's poses after the frame against the reference's
                    -> <= 1e-3 x max(1, extent); a frame above that must be ATTRIBUTED: the reference's own bundle adjustment, fed
                    OUR targets / weights for the same edges, must land on OUR poses (attr_dist within the same bound) -- i.e. the
                    frame's difference is exactly what the reference's solver makes of update-operator differences that are inside
                    their asserted tolerances;
  * and in the WELL-CONDITIONED scenario (the flow head scaled down and given a coherent image-wide shift, which bundle adjustment
    explains with the poses: tests/ref_harness.py:build_pair) every frame sits under the plain 1e-3, teacher forced AND free running.
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


K_NOISE = 3.0                   # a distance may reach K x the frame's noise floor where that exceeds the plain tolerance
WELL = dict(delta_scale=0.1, delta_bias=(1.0, -0.7))        # the well-conditioned teacher-forced scenario (see build_pair)
WELL_FREE = dict(delta_scale=0.01)                          # ... and the free-running one


def _poses_attributed(recs, tol=POSE_TOL, k=K_NOISE):
    """module docstring, POSES.  Per frame: lim0 = tol x max(1, extent); noise floor nf = max(yard, |reference - f64 solve|) -- how far
    the reference's own result is from itself re-run and from the exact step; lim = max(lim0, k x nf).
      * our BA on the reference's inputs:  |ours - reference| <= lim0, or |ours - f64 solve| <= lim;
      * the frame's poses:                 |ours - reference| <= lim0, or the difference is reproduced from OUR update outputs by the
                                           reference's solver or by the f64 solver (attr_dist / attr_exact <= lim).
    Returns the frames that needed more than the plain tolerance (for the printed report)."""
    special, bad = [], []
    for r in recs:
        if "pose_max" not in r:
            continue
        lim0 = tol * max(1.0, r.get("extent", 0.0))
        nf = max(r.get("yard", 0.0), r.get("ref_exact", 0.0))
        lim = max(lim0, k * nf)
        if r.get("ba_dist", 0.0) > lim0 and not min(r["ba_dist"], r.get("ours_exact", float("inf"))) <= lim:
            bad.append(("our BA on the reference's inputs", {k_: r.get(k_) for k_ in ("t", "ba_dist", "ours_exact", "ref_exact", "yard", "extent")}))
        if r["pose_max"] > lim0:
            attr = min(r.get("attr_dist") if r.get("attr_dist") is not None else float("inf"), r.get("attr_exact", float("inf")))
            if "ba_dist" in r and r["pose_max"] > lim and not attr <= lim:
                bad.append(("pose difference not reproduced from our update outputs", {k_: r.get(k_) for k_ in
                            ("t", "pose_max", "attr_dist", "attr_exact", "ref_exact", "yard", "extent")}))
        if max(r["pose_max"], r.get("ba_dist", 0.0)) > lim0:
            special.append({k_: (float(f"{r[k_]:.3g}") if isinstance(r.get(k_), float) else r.get(k_)) for k_ in
                            ("t", "pose_max", "extent", "yard", "ref_exact", "ba_dist", "ours_exact", "attr_dist", "attr_exact")})
    assert not bad, bad
    return special


def _report(name, recs, s, sing, n):
    at = [r for r in recs if "ba_dist" in r]
    n_pose = sum(1 for r in recs if "pose_max" in r and r["pose_max"] > POSE_TOL * max(1.0, r.get("extent", 0.0)))
    print(f"\n{name}: |pose| <= {POSE_TOL} x max(1, extent) on {n - n_pose}/{n} frames (median {np.median([r['pose_max'] for r in recs if 'pose_max' in r]):.2e}); "
          f"over {len(at)} bundle adjustments: reference re-run spread <= {s.get('yard_max', 0):.2e}, our BA on its inputs <= {s.get('ba_dist_max', 0):.2e}, "
          f"its BA on our targets vs our poses <= {(s.get('attr_dist_max') or 0):.2e}; hidden state max {s['net_max']:.2e} rms {s['net_rms']:.2e}; "
          f"target {s['target_max']:.2e} px, weight {s['weight_max']:.2e}; flow {s['flow_absdiff_max']:.2e} px; "
          f"depth rel. p50 {s['depth_rel_p50']:.2e} p90 {s['depth_rel_p90']:.2e}; frames beyond the plain tolerance: {sing}")


def _update_outputs_within_tolerance(s, scale=1.0):
    # the update operator's outputs against the reference's (f16 GEMMs on both sides, f16 correlation accumulate on the reference's):
    # hidden state 2e-2 (f16 ulp at |net| ~ 8), BA targets 2e-2 px, confidence weights 2e-3 -- measured 6e-3 / 8e-3 / 7e-4
    assert s["net_max"] < 2e-2 * scale and s["net_rms"] < 2e-3 * scale and s["target_max"] < 2e-2 * scale and s["weight_max"] < 2e-3 * scale


def test_teacher_forced_bench_configuration(dev, RP, stream):
    """every frame of an 80-frame run at the bench configuration as a one-step comparison (see module docstring): the STRESS case,
    random flow head on a static scene, run-away around frame 30"""
    frames, intr = stream
    ours, theirs, _ = H.build_pair(dev, HT, WD, M, KEYFRAME_THRESH=-1.0)
    recs = H.run_lockstep(ours, theirs, frames, 80, intr, feed=True, teacher=True, attribute_ba=True)
    s = _int_exact(recs, 80)
    sing = _poses_attributed(recs)
    _report("teacher forced (stress)", recs, s, sing, 80)
    assert s["E_last"] == 45312
    assert s["flow_absdiff_max"] < FLOW_TOL
    _update_outputs_within_tolerance(s)
    assert s["depth_rel_p50"] < 5e-3


def test_teacher_forced_well_conditioned(dev, RP, stream):
    """the same 80 frames, same arithmetic, with the flow head in a bounded regime (WELL): EVERY frame under the plain tolerance --
    1e-3 on every pose component, no yard-stick, no attribution needed (they are still measured and asserted)"""
    frames, intr = stream
    ours, theirs, _ = H.build_pair(dev, HT, WD, M, KEYFRAME_THRESH=-1.0, **WELL)
    recs = H.run_lockstep(ours, theirs, frames, 80, intr, feed=True, teacher=True, attribute_ba=True)
    s = _int_exact(recs, 80)
    sing = _poses_attributed(recs)
    _report("teacher forced (well conditioned)", recs, s, sing, 80)
    assert s["E_last"] == 45312 and not sing, sing
    assert s["pose_max"] < POSE_TOL
    assert s["flow_absdiff_max"] < FLOW_TOL
    _update_outputs_within_tolerance(s)


def test_free_running_well_conditioned(dev, RP, stream):
    """... and WITHOUT teacher forcing: both trackers free running for 70 frames in the bounded regime -- accumulated pose distance
    under 1e-3 x max(1, extent) on every frame (north_star: 'ATE within 1e-3 m of reference'), integer state bit-exact"""
    frames, intr = stream
    ours, theirs, _ = H.build_pair(dev, HT, WD, M, KEYFRAME_THRESH=-1.0, **WELL_FREE)
    recs = H.run_lockstep(ours, theirs, frames, 70, intr, feed=True)
    s = _int_exact(recs, 70)
    worst = max((r["pose_max"] / max(1.0, r["extent"]) for r in recs if "pose_max" in r), default=0.0)
    po, _ = ours.terminate()
    with torch.no_grad():
        pr, _ = theirs.terminate()
    raw, ali = H.trajectory_ate(po, pr)
    print(f"\nfree running (well conditioned): integer state bit-exact on 70/70 frames; accumulated pose distance <= {s['pose_max']:.2e} on a "
          f"trajectory of extent {s['extent_last']:.3g} (worst relative {worst:.2e}); after terminate(): ATE raw {raw:.2e}, Sim3-aligned {ali}")
    assert s["E_last"] == 45312 and worst < POSE_TOL
    assert raw < POSE_TOL * max(1.0, s["extent_last"])


def test_teacher_forced_end_to_end_encoders(dev, RP, stream):
    """as above, but the reference also runs its OWN encoders (torch / MIOpen convolutions under autocast) instead of being fed ours:
    the stated difference is the encoders' f16 arithmetic (tests/test_gpu_encoders.py: a few f16 ulps per feature)"""
    frames, intr = stream
    ours, theirs, _ = H.build_pair(dev, HT, WD, M, feed=False, KEYFRAME_THRESH=-1.0)
    recs = H.run_lockstep(ours, theirs, frames, 60, intr, feed=False, teacher=True, attribute_ba=True)
    s = _int_exact(recs, 60)
    sing = _poses_attributed(recs)
    _report("end to end", recs, s, sing, 60)
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
    recs = H.run_lockstep(ours, theirs, frames, 70, intr, feed=True, teacher=True, attribute_ba=True)
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
    # (a frame whose keyframe was dropped has no attribution against our final poses -- rings moved, edges renumbered --; such a
    #  frame is held to max(tolerance, K x yard) directly)
    _poses_attributed(recs)


def test_loop_closure_configuration(dev, RP, stream):
    """BASELINE config 5 (LOOP_CLOSURE=True): loop edges from PatchGraph.edges_loop (thresholded + NMS'd flow magnitudes), edges kept
    alive by the lc rule of dpvo.py:307-308, global BA over active + inactive edges (dpvo.py:312-326, EfficentE on the reference's
    side) -- 85 frames, teacher forced"""
    frames, intr = stream
    ours, theirs, _ = H.build_pair(dev, HT, WD, M, buffer=512, LOOP_CLOSURE=True, KEYFRAME_THRESH=-1.0)
    recs = H.run_lockstep(ours, theirs, frames, 85, intr, feed=True, teacher=True, attribute_ba=True)
    s = _int_exact(recs, 85)
    sing = _poses_attributed(recs)
    n_gba = sum(1 for r in recs if r.get("eff_impl"))
    gba_max = max((r["ba_dist"] for r in recs if r.get("eff_impl")), default=0.0)
    gb_o, gb_r = int(ours.ran_global_ba.sum()), int(theirs.ran_global_ba.sum())
    _report("loop closure", recs, s, sing, 85)
    print(f"loop closure: integer state (incl. loop edges) bit-exact on 85/85 frames, {gb_r} global BA runs on each side ({n_gba} of them the "
          f"frame's last BA: ours on EfficentE's inputs <= {gba_max:.2e}), {int(theirs.pg.ii_inac.numel())} inactive edges")
    assert gb_o == gb_r >= 2
    assert int((theirs.pg.jj - theirs.pg.ii > 30).sum()) > 0 or int(theirs.pg.ii_inac.numel()) > 0

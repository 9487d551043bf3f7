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

What is asserted, and why a failure can only have an implementation reason (round 5; round 4 allowed "at most two bad frames", a
constant tuned on a handful of runs, and went red on the driver's box):

  * the INTEGER state (frame / patch counters, edge lists, inactive lists, timestamps, patch coordinates, colours): bit-exact on EVERY
    frame of every scenario -- it does not depend on the conditioning of anything;

  * the FLOAT state, frame by frame under teacher forcing (after each frame our poses / depths / hidden state are reset to the
    reference's, so every frame is a one-step comparison at up to E = 45 312 edges), on every REGULAR frame:
      - the update operator's outputs (hidden state, BA targets, confidence weights) and the keyframe flow test's input within stated
        f16-level tolerances;
      - OUR bundle adjustment run on the reference's captured inputs against the reference's result: <= 1e-3 x max(1, extent);
      - our poses after the frame against the reference's: <= 1e-3 x max(1, extent) -- or, if not, the difference must be ATTRIBUTED:
        the reference's own BA (or the f64 solve) fed OUR targets / weights lands on OUR poses within the same bound, i.e. the frame's
        difference is what the reference's own solver makes of update-operator differences that are inside their asserted tolerances.

  * what a REGULAR frame is, is decided by the REFERENCE ALONE, never by our result (tests/ref_harness.py:attribute): every frame's BA
    call of the reference is captured and re-run three times on its own inputs (yard: the reference-vs-reference one-step spread on
    this box; float atomics are its only source of difference), and, where a distance exceeds the plain tolerance, solved in f64 by
    the CPU oracle (ref_exact: the reference's distance to the exact step).  A frame whose noise floor max(yard, ref_exact) exceeds
    1e-3 x max(1, extent) is SINGULAR: the reference cannot reproduce its own step there to the tolerance we are held to (measured:
    yard up to 6e-1, and at such frames |ours - f64| / |reference - f64| came out anywhere between 0.08 and 9 -- no factor K on a
    singular frame's spread is a sound assertion).  A tracker with random weights on a static scene gets there by construction: zero
    baseline makes depth unobservable (C = sum w (Jp . t_ij)^2 -> 0, the depth step is u / (C + 1e-4)), the flow head emits noise, and
    around frame 30 the scale runs away.  The state a singular step leaves behind is degenerate (points at the Z clamp: reprojections
    differ by pixels between any two f32 implementations), so the float assertions of a run END at its first singular frame; the
    integer assertions go on to the last frame, and every scenario states how many regular frames it must at least have had.

  * the BOUNDED scenario (WELL: the flow head's last layer x 0.003 -- the SAME arithmetic on every edge, with updates small enough
    that the run-away does not happen within the run): all 80 teacher-forced frames regular and under the plain 1e-3, E = 45 312 from
    frame 44 on; and FREE RUNNING (no teacher forcing) the accumulated pose distance stays under 1e-3 absolute AND under 2e-3 of the
    trajectory's extent (measured 4e-7 / 4e-4), with the final ATE after terminate() reported.

  * STEP-RELATIVE bounds (round 6): an absolute 1e-3 says nothing where the whole step is 2e-5 -- a bundle
    adjustment that did NOTHING would pass it.  Every one-step float assertion is therefore ALSO held to the size of the reference's
    own step that frame (tests/ref_harness.py: step_ref = |poses_after - poses_before| of its captured BA call, step_frame = the sum
    over the frame's calls): our BA on the reference's inputs <= max(5 x noise floor, 0.05 x step_ref), our poses after the frame <=
    max(5 x noise floor, 0.1 x step_frame), noise floor = max(yard, ref_exact) as above.  Measured over flow-head scales 0.003 .. 0.3
    (profiles/r06_a_delta_scale_sweep.txt): |BA difference| / step <= 1e-2, |pose difference| / step <= 3.5e-2 (the initialisation
    frame; <= 1e-2 elsewhere) -- the pose figure contains the update operators' f16-level differences in targets / weights, so it is
    not 1e-4; a skipped or truncated bundle adjustment sits at 1.0.  test_negative_control_* runs OUR tracker with its bundle
    adjustment replaced by a no-op and requires these checks to flag every frame.

  * the MID-SCALE scenario (MID: flow head x 0.3): the reference stays regular on all but one or two of its 80 teacher-forced frames
    and the trajectory's extent reaches ~10 (>= 0.05 required); see test_teacher_forced_mid_scale for why not 0.1.  Free running at that scale is NOT a test: the
    reference run against ITSELF (two instances, same inputs; float atomics only) is 0.10 apart at an extent of 0.96 by frame 76 at
    scale 0.03 already (profiles/r06_b_ref_vs_ref_free_running.txt) -- a chaotic recurrence whatever implements it.
The tolerances are written where they are asserted; the measured values are printed (-s) and committed under profiles/."""
import numpy as np
import pytest
import torch

from tests import ref_harness as H

pytestmark = pytest.mark.gpu

HT, WD, M = 480, 640, 96        # BASELINE config 2: what bench.py times
POSE_TOL = 1e-3                 # north_star: "ATE within 1e-3 m of reference"; applied to every pose component, every regular frame
FLOW_TOL = 1e-3                 # px, keyframe flow test input (dpvo.py:257-270) under teacher forcing (measured 3-8e-5 with equal poses)
# ... plus what the frame's own pose difference moves a reprojected pixel by: the flow test runs AFTER the frame's BA, on poses that differ
# by r["pose_max"] (bounded separately below); at 1/4 resolution fx = 80 px, inverse depths ~ 1, two poses per edge -> 160 px per unit pose.
# (Without it the full-scale scenarios fail one run in eight: 1.2e-3 px at a frame whose poses differ by the reference's own 1e-5 noise.)
FLOW_PER_POSE = 160.0
# the update operator's outputs against the reference's (f16 GEMMs on both sides, f16 correlation accumulate on the reference's):
# hidden state 2e-2 (f16 ulp at |net| ~ 8), rms 2e-3, BA targets 2e-2 px, confidence weights 2e-3 -- measured 8e-3 / 6e-4 / 1.2e-2 / 1e-3
OUT_TOL = dict(net_max=2e-2, net_rms=2e-3, target_max=2e-2, weight_max=2e-3)
WELL = dict(delta_scale=0.003)  # the bounded scenario (tests/ref_harness.py:build_pair)
MID = dict(delta_scale=0.3)     # the mid-scale scenario: the tracker leaves its start around frame 37 and then moves 0.1-0.5 per frame (extent ~10)
# step-relative bounds (module docstring): K_NOISE x the reference's own noise floor, or a fraction of the reference's own step
K_NOISE, BA_STEP, POSE_STEP = 5.0, 0.05, 0.1
FAST = dict(REMOVAL_WINDOW=16, OPTIMIZATION_WINDOW=7, PATCH_LIFETIME=11)      # config/fast.yaml:4-7 (+ PATCHES_PER_FRAME = 48)


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


def _float_state(name, recs, min_regular, tol=POSE_TOL, out_scale=1.0, flow_tol=FLOW_TOL, collect=False, skip_singular=False):
    """module docstring, FLOAT state: strict assertions on every frame up to the run's first singular frame (decided by the reference's
    own noise floor); at least `min_regular` frames must have been regular.  Prints the measured values; returns the first singular
    frame (None: the whole run was regular).  collect=True: returns (first singular frame, list of violations) instead of asserting.
    skip_singular=True (teacher-forced runs only: every frame starts from the REFERENCE's state, so a frame after a singular one is a
    sound one-step comparison again whenever the reference's own noise floor says so): singular frames are skipped, not terminal."""
    first, bad, checked, attributed = None, [], 0, []
    worst = dict(pose=0.0, ba=0.0, net_max=0.0, net_rms=0.0, target_max=0.0, weight_max=0.0, flow=0.0, yard=0.0, ba_rel=0.0, pose_rel=0.0,
                 step_min=float("inf"), step_max=0.0)
    for r in recs:
        if "pose_max" not in r:
            continue
        t, lim0 = r["t"], tol * max(1.0, r.get("extent", 0.0))
        # the update operator ran on a regular state: its outputs are held to their tolerances whatever the BA then makes of them
        for k, v in OUT_TOL.items():
            if k in r:
                worst[k] = max(worst[k], r[k])
                if r[k] >= v * out_scale:
                    bad.append((t, k, r[k], v * out_scale))
        if r.get("flow_ours") is not None and r.get("flow_ref") is not None and r["flow_ours"] == r["flow_ours"]:
            worst["flow"] = max(worst["flow"], abs(r["flow_ours"] - r["flow_ref"]))
            if abs(r["flow_ours"] - r["flow_ref"]) >= flow_tol + FLOW_PER_POSE * r["pose_max"]:
                bad.append((t, "flow", r["flow_ours"], r["flow_ref"], r["pose_max"]))
        nf = max(r.get("yard", 0.0), r.get("ref_exact", 0.0))
        if nf > lim0:
            first = t if first is None else first
            if skip_singular:
                continue
            break
        checked += 1
        # the step-relative bounds (only where the frame's BA was captured: step_ref is the reference's own step)
        lim_ba = lim_pose = lim0
        if "step_ref" in r:
            lim_ba = min(lim0, max(K_NOISE * nf, BA_STEP * r["step_ref"]))
            lim_pose = min(lim0, max(K_NOISE * nf, POSE_STEP * r.get("step_frame", r["step_ref"])))
            worst["step_min"], worst["step_max"] = min(worst["step_min"], r["step_ref"]), max(worst["step_max"], r["step_ref"])
            if r["step_ref"] > 0:
                worst["ba_rel"] = max(worst["ba_rel"], r.get("ba_dist", 0.0) / r["step_ref"])
                worst["pose_rel"] = max(worst["pose_rel"], r["pose_max"] / max(r.get("step_frame", 0.0), r["step_ref"]))
        worst["yard"] = max(worst["yard"], r.get("yard", 0.0))
        worst["ba"] = max(worst["ba"], r.get("ba_dist", 0.0))
        worst["pose"] = max(worst["pose"], r["pose_max"] / max(1.0, r.get("extent", 0.0)))
        if r.get("ba_dist", 0.0) > lim_ba and not r.get("ours_exact", float("inf")) <= lim_ba:
            bad.append((t, "our BA on the reference's inputs", {k: r.get(k) for k in ("ba_dist", "step_ref", "ours_exact", "ref_exact", "yard", "extent")}))
        if r["pose_max"] > lim_pose:
            attr = min(r.get("attr_dist") if r.get("attr_dist") is not None else float("inf"), r.get("attr_exact", float("inf")))
            attributed.append((t, float(f"{r['pose_max']:.3g}"), float(f"{attr:.3g}")))
            if not attr <= lim_pose:
                bad.append((t, "pose difference not reproduced from our update outputs",
                            {k: r.get(k) for k in ("pose_max", "step_ref", "step_frame", "attr_dist", "attr_exact", "ref_exact", "yard", "extent")}))
    sing = [{k: (float(f"{r[k]:.3g}") if isinstance(r.get(k), float) else r.get(k)) for k in
             ("t", "pose_max", "extent", "yard", "ref_exact", "ba_dist", "ours_exact", "attr_dist", "attr_exact")}
            for r in recs if "pose_max" in r and max(r.get("yard", 0.0), r.get("ref_exact", 0.0)) > tol * max(1.0, r.get("extent", 0.0))]
    n_ba = sum(1 for r in recs if "ba_dist" in r)
    print(f"\n{name}: {checked} regular frames checked ({n_ba} bundle adjustments captured), first singular frame {first}; on the regular frames: "
          f"|pose| / max(1, extent) <= {worst['pose']:.2e}, our BA on the reference's inputs <= {worst['ba']:.2e} (the reference re-run on them <= "
          f"{worst['yard']:.2e}); the reference's own step {worst['step_min']:.2e} .. {worst['step_max']:.2e}, |our BA - its BA| / step <= "
          f"{worst['ba_rel']:.2e}, |pose difference| / step <= {worst['pose_rel']:.2e}; hidden state max {worst['net_max']:.2e} rms "
          f"{worst['net_rms']:.2e}, target {worst['target_max']:.2e} px, weight "
          f"{worst['weight_max']:.2e}, flow {worst['flow']:.2e} px; frames that needed the attribution (t, |pose|, attributed to within): {attributed}; "
          f"singular frames of the whole run (reported, not asserted): {sing[:6]}")
    if collect:
        return first, bad
    assert not bad, bad
    assert checked >= min_regular, f"only {checked} regular frames before the first singular one (t = {first}); {min_regular} required"
    return first


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
    # (t < 16: the chaotic amplification sets in between frames 20 and 30 and its onset moves with the reference's own float-atomics
    #  noise from run to run -- measured over six runs: 4e-5 .. 5e-4 up to t = 28, 2.4e-4 .. 5.4e-4 once at t = 20, never above 7e-5 up
    #  to t = 16.  The whole-run free-running pose assertion lives in test_free_running_bounded)
    assert s["pose_max_first16"] < POSE_TOL
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
    """the STRESS case: random flow head at full scale on a static scene, every frame a one-step comparison until the scale runs away
    (first singular frame: 30 .. 33 in every run so far); integer state bit-exact on all 80 frames"""
    frames, intr = stream
    ours, theirs, _ = H.build_pair(dev, HT, WD, M, KEYFRAME_THRESH=-1.0)
    recs = H.run_lockstep(ours, theirs, frames, 80, intr, feed=True, teacher=True, attribute_ba=True)
    s = _int_exact(recs, 80)
    _float_state("teacher forced (stress)", recs, min_regular=18)       # (t = 8 .. 25 at least: measured 22-25 regular frames)
    assert s["E_last"] == 45312


def test_teacher_forced_bounded(dev, RP, stream):
    """the same 80 frames, same arithmetic, flow head x 0.003 (WELL): EVERY frame regular and under the plain tolerance, 36 of them at
    E = 45 312"""
    frames, intr = stream
    ours, theirs, _ = H.build_pair(dev, HT, WD, M, KEYFRAME_THRESH=-1.0, **WELL)
    recs = H.run_lockstep(ours, theirs, frames, 80, intr, feed=True, teacher=True, attribute_ba=True)
    s = _int_exact(recs, 80)
    first = _float_state("teacher forced (bounded)", recs, min_regular=72)
    assert first is None and s["E_last"] == 45312, f"a singular frame (t = {first}) in the bounded scenario"
    assert s["pose_max"] < POSE_TOL


def test_free_running_bounded(dev, RP, stream):
    """... and WITHOUT teacher forcing: both trackers free running for 64 frames in the bounded regime -- the ACCUMULATED pose distance
    under 1e-3 absolute (north_star: 'ATE within 1e-3 m of reference') and under 2e-3 of the trajectory's extent on every frame
    (measured 4e-7 and 4e-4); integer state bit-exact; ATE after terminate() (12 more updates each) under 1e-3"""
    frames, intr = stream
    n = 64
    ours, theirs, _ = H.build_pair(dev, HT, WD, M, KEYFRAME_THRESH=-1.0, **WELL)
    recs = H.run_lockstep(ours, theirs, frames, n, intr, feed=True)
    s = _int_exact(recs, n)
    rel = max((r["pose_max"] / r["extent"] for r in recs if r.get("extent", 0.0) > 0 and r["t"] >= 12), default=0.0)
    po, _ = ours.terminate()
    with torch.no_grad():
        pr, _ = theirs.terminate()
    raw, ali = H.trajectory_ate(po, pr)
    print(f"\nfree running (bounded): integer state bit-exact on {n}/{n} frames; accumulated pose distance <= {s['pose_max']:.2e} on a trajectory "
          f"of extent {s['extent_last']:.3g} (worst distance / extent {rel:.2e}); after terminate(): ATE raw {raw:.2e}, Sim3-aligned {ali}")
    assert s["E_last"] == 45312
    assert s["pose_max"] < POSE_TOL and rel < 2e-3
    # (an absolute 1e-3 on a trajectory of that extent is no statement; the ATE relative to the extent is)
    assert raw < POSE_TOL and raw <= 1e-3 * s["extent_last"], (raw, s["extent_last"])


def test_teacher_forced_mid_scale(dev, RP, stream):
    """MID (flow head x 0.3): a trajectory with an extent worth the name.  At the swept mid scales the random-weight tracker sits still for
    a while (extent 0.006-0.04) and then leaves in one or two steps of 0.05-0.4 (profiles/r06_a_delta_scale_sweep.txt) -- at scale 0.1
    anywhere between frame 50 and never within 80 frames from run to run (the reference's float atomics decide), at 0.3 around frame 37;
    from there on every frame is a step of 0.1-0.5 on an extent of 1-10 that the reference reproduces to 1e-6-1e-4: the frames a
    one-step comparison is worth most on.  The leaving frame itself can be singular by the reference's own account (yard 2e-3, or its
    f32 result 1.9e-3 away from the f64 solution where ours is 3.8e-5 away) -- so this test SKIPS singular frames instead of ending at
    the first one (teacher forcing restarts every frame from the reference's state) and requires: >= 60 regular frames, >= 5 of them at
    an extent >= 0.05 with a reference step >= 0.01, the absolute and the step-relative bounds on every regular frame."""
    frames, intr = stream
    ours, theirs, _ = H.build_pair(dev, HT, WD, M, KEYFRAME_THRESH=-1.0, **MID)
    recs = H.run_lockstep(ours, theirs, frames, 80, intr, feed=True, teacher=True, attribute_ba=True)
    s = _int_exact(recs, 80)
    first = _float_state("teacher forced (mid scale)", recs, min_regular=60, skip_singular=True)
    big = [r for r in recs if "step_ref" in r and max(r.get("yard", 0.0), r.get("ref_exact", 0.0)) <= POSE_TOL * max(1.0, r["extent"])
           and r["extent"] >= 0.05 and r["step_ref"] >= 0.01]
    print(f"mid scale: extent {s['extent_last']:.3g}, first singular frame {first}, regular frames at extent >= 0.05 with a step >= 0.01: {len(big)}"
          + (f" (steps {min(r['step_ref'] for r in big):.3g} .. {max(r['step_ref'] for r in big):.3g}, |pose difference| / step <= "
             f"{max(r['pose_max'] / r['step_ref'] for r in big):.2e})" if big else ""))
    assert s["E_last"] == 45312 and s["extent_last"] >= 0.05 and len(big) >= 5


def test_negative_control_noop_bundle_adjustment(dev, RP, stream, monkeypatch):
    """The checker must be able to FAIL: our tracker with dpvo_amd.fastba.BA replaced by a no-op (call-by-call path, so that the Python
    entry is the one that runs; the same entry is what the harness calls for `our BA on the reference's inputs`) in the bounded
    scenario -- where an absolute 1e-3 would pass a tracker that never adjusts anything.  Required: the step-relative bounds flag
    BOTH quantities on every frame whose BA was captured, and the integer state stays exact (it does not depend on the floats)."""
    from dpvo_amd import dpvo as dpvo_mod, fastba as our_fastba
    frames, intr = stream
    monkeypatch.setattr(dpvo_mod, "_FRAME_CALL", False)
    monkeypatch.setattr(our_fastba, "BA", lambda *a, **k: [])
    ours, theirs, _ = H.build_pair(dev, HT, WD, M, KEYFRAME_THRESH=-1.0, **WELL)
    recs = H.run_lockstep(ours, theirs, frames, 30, intr, feed=True, teacher=True, attribute_ba=True)
    _int_exact(recs, 30)
    assert ours._fu is None
    first, bad = _float_state("negative control (no-op BA)", recs, min_regular=0, collect=True)
    n_ba = sum(1 for r in recs if "ba_dist" in r)
    flagged_ba = {b[0] for b in bad if b[1] == "our BA on the reference's inputs"}
    flagged_pose = {b[0] for b in bad if b[1] == "pose difference not reproduced from our update outputs"}
    print(f"negative control: {n_ba} captured frames, BA check flagged {len(flagged_ba)}, pose check flagged {len(flagged_pose)}; "
          f"under the absolute bound alone: {sum(1 for r in recs if r.get('ba_dist', 0.0) > POSE_TOL * max(1.0, r.get('extent', 0.0)))} frames")
    assert first is None and n_ba >= 20
    assert len(flagged_ba) == n_ba and len(flagged_pose) >= n_ba - 1, (n_ba, sorted(flagged_ba), sorted(flagged_pose))


def test_fast_yaml_lockstep(dev, RP, stream):
    """config/fast.yaml:1-19 (48 patches per frame, REMOVAL_WINDOW 16, OPTIMIZATION_WINDOW 7, PATCH_LIFETIME 11): other plan windows, tile
    counts and M than every other tracker-level test.  Steady state E = 48 (16 x 11 + sum_{a<16} min(10, a)) = 48 (176 + 105) = 13 488
    on BOTH trackers (SURVEY appendix A.1 prints 48 (176 + 95) = 13 008: the second sum is 0 + 1 + ... + 9 + 6 x 10 = 105).  44 frames
    teacher forced in the bounded scenario: integer state bit-exact, every frame regular, absolute and step-relative bounds; and the
    one-call frame path is the one that ran."""
    frames, intr = stream
    ours, theirs, _ = H.build_pair(dev, HT, WD, 48, KEYFRAME_THRESH=-1.0, **FAST, **WELL)
    recs = H.run_lockstep(ours, theirs, frames, 44, intr, feed=True, teacher=True, attribute_ba=True)
    s = _int_exact(recs, 44)
    first = _float_state("fast.yaml (bounded)", recs, min_regular=30)
    assert first is None and s["E_last"] == 13488 and ours._fu is not None, (first, s["E_last"])


def test_teacher_forced_end_to_end_encoders(dev, RP, stream):
    """bounded scenario, but the reference also runs its OWN encoders (torch / MIOpen convolutions under autocast) instead of being fed
    ours: the stated difference is the encoders' f16 arithmetic (tests/test_gpu_encoders.py: a few f16 ulps per feature), so the update
    operator's outputs get twice the tolerance and the flow test's input 2e-2 px (flow magnitudes differ with the features)"""
    frames, intr = stream
    ours, theirs, _ = H.build_pair(dev, HT, WD, M, feed=False, KEYFRAME_THRESH=-1.0, **WELL)
    recs = H.run_lockstep(ours, theirs, frames, 60, intr, feed=False, teacher=True, attribute_ba=True)
    _int_exact(recs, 60)
    first = _float_state("end to end (bounded)", recs, min_regular=50, out_scale=2.0, flow_tol=2e-2)
    assert first is None


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
    # A decision may differ only on a knife edge: the reference's flow within the flow tolerance (FLOW_TOL + FLOW_PER_POSE x the frame's pose difference) of the threshold (its own flows move by more than
    # that between two of its runs; ours differ from them by 3-6e-5 px).  The run stops at such a frame -- the integer states part
    # there by definition -- and everything before it must be exact.  (Six runs so far: no such frame, smallest margin 1.2e-3 px.)
    bad = s["first_decision_mismatch"]
    edge = FLOW_TOL + FLOW_PER_POSE * max((r.get("pose_max", 0.0) for r in recs if bad is not None and r["t"] == bad["t"]), default=0.0)
    assert bad is None or abs(bad["flow_ref"] - thr) < edge, (bad, edge)
    ok_frames = len(recs) if bad is None else len(recs) - 1
    assert s["int_equal_frames"] >= ok_frames and (bad is not None or s["int_equal_frames"] == 70), s["first_int_mismatch"]
    assert len(dec) >= 40 and drops >= 10 and len(dec) - drops >= 10, "both branches of dpvo.py:272 must be exercised"
    # (full-scale flow head: with keyframes dropped the window never fills with near-static frames and the scale run-away comes late or
    #  not at all -- measured: no singular frame in 70; the float assertions still end at the first one if there is one)
    _float_state("unscripted decisions", recs, min_regular=18)


def test_loop_closure_configuration(dev, RP, stream):
    """BASELINE config 5 (LOOP_CLOSURE=True): loop edges from PatchGraph.edges_loop (thresholded + NMS'd flow magnitudes), edges kept
    alive by the lc rule of dpvo.py:307-308, global BA over active + inactive edges (dpvo.py:312-326, EfficentE on the reference's
    side) -- 85 frames, teacher forced, bounded scenario: every frame regular, every global BA of ours compared with EfficentE's on the
    same inputs"""
    frames, intr = stream
    ours, theirs, _ = H.build_pair(dev, HT, WD, M, buffer=512, LOOP_CLOSURE=True, KEYFRAME_THRESH=-1.0, **WELL)
    recs = H.run_lockstep(ours, theirs, frames, 85, intr, feed=True, teacher=True, attribute_ba=True)
    s = _int_exact(recs, 85)
    first = _float_state("loop closure (bounded)", recs, min_regular=70)
    n_gba = sum(1 for r in recs if r.get("eff_impl"))
    gba_max = max((r["ba_dist"] for r in recs if r.get("eff_impl")), default=0.0)
    gb_o, gb_r = int(ours.ran_global_ba.sum()), int(theirs.ran_global_ba.sum())
    print(f"loop closure: integer state (incl. loop edges) bit-exact on 85/85 frames, {gb_r} global BA runs on each side ({n_gba} of them the "
          f"frame's last BA: ours on EfficentE's inputs <= {gba_max:.2e}), {int(theirs.pg.ii_inac.numel())} inactive edges, first singular frame {first}")
    assert gb_o == gb_r >= 2 and n_gba >= 2
    assert int((theirs.pg.jj - theirs.pg.ii > 30).sum()) > 0 or int(theirs.pg.ii_inac.numel()) > 0


def test_loop_closure_mid_scale(dev, RP, stream):
    """config 5 at the mid scale (the bounded LOOP_CLOSURE run compares the global BA -- block-sparse linearisation, device
    Cholesky, wide plans -- with EfficentE only on steps of ~1e-5, and the full-scale run ends before the first global BA fires): flow
    head x 0.3, teacher forced, singular frames skipped -- the global BAs of the frames on which the tracker MOVES (steps 0.05-1.4).
    Required: integer state incl. loop edges bit-exact on 85 frames, the same global-BA triggers, and at least three REGULAR frames
    whose last BA was a global one with a reference step >= 1e-3, all within the absolute and step-relative bounds."""
    frames, intr = stream
    ours, theirs, _ = H.build_pair(dev, HT, WD, M, buffer=512, LOOP_CLOSURE=True, KEYFRAME_THRESH=-1.0, **MID)
    recs = H.run_lockstep(ours, theirs, frames, 85, intr, feed=True, teacher=True, attribute_ba=True)
    _int_exact(recs, 85)
    first = _float_state("loop closure (mid scale)", recs, min_regular=60, skip_singular=True)
    reg = lambda r: max(r.get("yard", 0.0), r.get("ref_exact", 0.0)) <= POSE_TOL * max(1.0, r.get("extent", 0.0))
    gba = [r for r in recs if r.get("eff_impl") and reg(r)]
    big = [r for r in gba if r["step_ref"] >= 1e-3]
    gb_o, gb_r = int(ours.ran_global_ba.sum()), int(theirs.ran_global_ba.sum())
    print(f"loop closure (mid scale): {gb_r} global BA runs on each side, {len(gba)} of them the last BA of a regular frame, {len(big)} with a "
          f"reference step >= 1e-3" + (f" (steps {min(r['step_ref'] for r in big):.3g} .. {max(r['step_ref'] for r in big):.3g}, |our BA - "
          f"EfficentE| / step <= {max(r['ba_dist'] / r['step_ref'] for r in big):.2e}, E up to {max(r['E_ba'] for r in big)})" if big else "")
          + f", first singular frame {first}")
    assert gb_o == gb_r >= 2 and len(big) >= 3


def test_loop_closure_stress(dev, RP, stream):
    """config 5 with the flow head at full scale (the run-away scenario): the INTEGER state -- which loop edges edges_loop selects,
    which edges the lc rule keeps alive, when the global BA triggers -- bit-exact over 85 frames whatever the floats do; float
    assertions until the first singular frame"""
    frames, intr = stream
    ours, theirs, _ = H.build_pair(dev, HT, WD, M, buffer=512, LOOP_CLOSURE=True, KEYFRAME_THRESH=-1.0)
    recs = H.run_lockstep(ours, theirs, frames, 85, intr, feed=True, teacher=True, attribute_ba=True)
    _int_exact(recs, 85)
    _float_state("loop closure (stress)", recs, min_regular=18)
    assert int(ours.ran_global_ba.sum()) == int(theirs.ran_global_ba.sum()) >= 2

#!/usr/bin/env python
"""Trajectory-level parity report: dpvo_amd's tracker against the REFERENCE's own tracker (its Python + its native kernels compiled
for gfx950, oracle/ref_pipeline.py) on the MI355X, at the configuration bench.py times (480x640, 96 patches, default.yaml, one C-ABI
call per frame, overlapped + held encoders, deferred result record).  Prints one JSON object per scenario; the committed copy is
profiles/rNN_ref_parity_pipeline.txt.  tests/test_zz_ref_pipeline.py asserts the same quantities.

    python tools/ref_parity.py [--frames 70] [--scenarios A,A2,B,C,D]
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tests import ref_harness as H              # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=70)
    ap.add_argument("--scenarios", default="A,A2,B,C,D")
    ap.add_argument("--ht", type=int, default=480)
    ap.add_argument("--wd", type=int, default=640)
    ap.add_argument("--patches", type=int, default=96)
    ap.add_argument("--delta-scale", type=float, default=1.0, help="scale of the flow head's last layer (see tests/ref_harness.py)")
    ap.add_argument("--delta-bias", default=None, help="bx,by: a coherent image-wide shift added to the flow head's bias (tests/ref_harness.py)")
    ap.add_argument("--attribute", action="store_true", help="teacher-forced scenarios: take every frame's BA apart (yard / ba_dist / attr_dist)")
    args = ap.parse_args()
    args.delta_bias = None if args.delta_bias is None else tuple(float(v) for v in args.delta_bias.split(","))
    from oracle import ref_pipeline as RP
    if not RP.available():
        print(json.dumps({"error": "oracle/_ref is not built (run oracle/build_ref.py where /root/reference exists)"}))
        return 1
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    ht, wd, M = args.ht, args.wd, args.patches
    frames = H.stream(64, ht, wd, dev)
    intr = torch.tensor([320.0 * wd / 640, 320.0 * wd / 640, wd / 2.0, ht / 2.0], device=dev)
    todo = args.scenarios.split(",")
    flows_A = None
    final_A = None

    def emit(name, what, recs, extra=None, t0=None):
        s = H.summarise(recs)
        s.update(scenario=name, what=what, **(extra or {}))
        if t0 is not None:
            s["seconds"] = round(time.perf_counter() - t0, 1)
        print(json.dumps(s), flush=True)
        return s

    if "A" in todo or "A2" in todo or "C" in todo:
        t0 = time.perf_counter()
        ours, theirs, cfg = H.build_pair(dev, ht, wd, M, delta_scale=args.delta_scale, delta_bias=args.delta_bias, KEYFRAME_THRESH=-1.0)
        recs = H.run_lockstep(ours, theirs, frames, args.frames, intr, feed=True)
        flows_A = [r["flow_ref"] for r in recs if r.get("flow_ref") is not None]
        final_A = RP.snapshot(ours)
        po, _ = ours.terminate()
        with torch.no_grad():
            pr, _ = theirs.terminate()
        raw, ali = H.trajectory_ate(po, pr)
        emit("A", "bench configuration (no keyframe dropped), encoders of dpvo_amd feed both trackers, one-call path + overlapped held "
                  "encoders + deferred record", recs,
             {"ate_raw_after_terminate": raw, "ate_sim3_after_terminate": ali, "flow_ref_min_med_max":
              [float(np.min(flows_A)), float(np.median(flows_A)), float(np.max(flows_A))] if flows_A else None}, t0)
        del ours, theirs
    if "A2" in todo and final_A is not None:
        t0 = time.perf_counter()
        ours, theirs, cfg = H.build_pair(dev, ht, wd, M, delta_scale=args.delta_scale, delta_bias=args.delta_bias, KEYFRAME_THRESH=-1.0)
        with torch.no_grad():
            for t in range(args.frames):
                torch.manual_seed(5000 + t)
                ours(float(t), frames[t % 64], intr, image_ready=False)       # no flush between frames: exactly what bench.py's loop does
            ours.flush()
        s2 = RP.snapshot(ours)
        same = all(np.array_equal(s2[k], final_A[k]) for k in ("ii", "jj", "kk", "poses", "patches"))
        print(json.dumps({"scenario": "A2", "what": "our tracker without a flush between frames (bench.py's loop) ends bit-identical to "
                          "the lock-step run of A", "bit_identical": bool(same), "seconds": round(time.perf_counter() - t0, 1)}), flush=True)
        del ours, theirs
    if "B" in todo:
        t0 = time.perf_counter()
        ours, theirs, cfg = H.build_pair(dev, ht, wd, M, delta_scale=args.delta_scale, delta_bias=args.delta_bias, feed=False, KEYFRAME_THRESH=-1.0)
        recs = H.run_lockstep(ours, theirs, frames, min(args.frames, 40), intr, feed=False)
        emit("B", "as A but fully end-to-end: the reference runs its own encoders (torch / MIOpen convolutions under autocast)", recs, None, t0)
        del ours, theirs
    if "C" in todo and flows_A:
        t0 = time.perf_counter()
        thr = float(np.median(flows_A))
        ours, theirs, cfg = H.build_pair(dev, ht, wd, M, delta_scale=args.delta_scale, delta_bias=args.delta_bias, KEYFRAME_THRESH=thr)
        recs = H.run_lockstep(ours, theirs, frames, args.frames, intr, feed=True, stop_on_mismatch=True)
        emit("C", f"unscripted keyframe decisions: KEYFRAME_THRESH = {thr:.4f} (median flow of scenario A), no override on either side", recs,
             {"thresh": thr, "decisions_list": [(r["t"], int(r["drop_ours"]), int(r["drop_ref"]), r["flow_ours"], r["flow_ref"])
                                                for r in recs if r.get("flow_ref") is not None]}, t0)
        del ours, theirs
    if "T" in todo:
        t0 = time.perf_counter()
        ours, theirs, cfg = H.build_pair(dev, ht, wd, M, delta_scale=args.delta_scale, delta_bias=args.delta_bias, KEYFRAME_THRESH=-1.0)
        recs = H.run_lockstep(ours, theirs, frames, args.frames, intr, feed=True, teacher=True, attribute_ba=args.attribute)
        emit("T", "as A with teacher forcing: after every frame our float state (poses, depths, hidden state) is reset to the reference's, "
                  "so each frame measures one frame's divergence at the bench configuration", recs, None, t0)
        del ours, theirs
    if "TB" in todo:
        t0 = time.perf_counter()
        ours, theirs, cfg = H.build_pair(dev, ht, wd, M, delta_scale=args.delta_scale, delta_bias=args.delta_bias, feed=False, KEYFRAME_THRESH=-1.0)
        recs = H.run_lockstep(ours, theirs, frames, args.frames, intr, feed=False, teacher=True, attribute_ba=args.attribute)
        emit("TB", "as T, fully end-to-end (the reference runs its own torch / MIOpen encoders)", recs, None, t0)
        del ours, theirs
    if "TD" in todo:
        t0 = time.perf_counter()
        ours, theirs, cfg = H.build_pair(dev, ht, wd, M, delta_scale=args.delta_scale, delta_bias=args.delta_bias, buffer=512, LOOP_CLOSURE=True, KEYFRAME_THRESH=-1.0)
        recs = H.run_lockstep(ours, theirs, frames, max(args.frames, 85), intr, feed=True, teacher=True, attribute_ba=args.attribute)
        emit("TD", "LOOP_CLOSURE=True (BASELINE config 5) with teacher forcing: loop edges + global BA on both sides", recs,
             {"global_ba_runs_ours": int(ours.ran_global_ba.sum()), "global_ba_runs_ref": int(theirs.ran_global_ba.sum()),
              "inactive_edges": int(theirs.pg.ii_inac.numel())}, t0)
        del ours, theirs
    if "F" in todo:
        t0 = time.perf_counter()
        ours, theirs, cfg = H.build_pair(dev, ht, wd, M, delta_scale=args.delta_scale, delta_bias=args.delta_bias, KEYFRAME_THRESH=-1.0, ref_over={"MIXED_PRECISION": False})
        recs = H.run_lockstep(ours, theirs, frames, args.frames, intr, feed=True)
        emit("F", "as A, but the reference runs with MIXED_PRECISION=False: f32 feature buffers, so its correlation kernel accumulates in "
                  "f32 like ours (the update operator stays under autocast, dpvo.py:332) -- isolates the reference's f16 correlation arithmetic",
             recs, None, t0)
        del ours, theirs
    if "R" in todo:
        t0 = time.perf_counter()
        ra, rb, cfg = H.build_pair(dev, ht, wd, M, delta_scale=args.delta_scale, delta_bias=args.delta_bias, feed=False, ours=False, KEYFRAME_THRESH=-1.0)
        recs = H.run_lockstep(ra, rb, frames, args.frames, intr, feed=False)
        emit("R", "the reference against ITSELF (two instances, same inputs): its own run-to-run distance (float atomics in BA, "
                  "ba_cuda.cu:335-373) amplified by the same dynamics -- the yard-stick for A", recs, None, t0)
        del ra, rb
    if "D" in todo:
        t0 = time.perf_counter()
        ours, theirs, cfg = H.build_pair(dev, ht, wd, M, delta_scale=args.delta_scale, delta_bias=args.delta_bias, buffer=512, LOOP_CLOSURE=True, KEYFRAME_THRESH=-1.0)
        recs = H.run_lockstep(ours, theirs, frames, max(args.frames, 85), intr, feed=True)
        emit("D", "LOOP_CLOSURE=True (BASELINE config 5): loop edges + global BA on both sides", recs,
             {"global_ba_runs_ours": int(ours.ran_global_ba.sum()), "global_ba_runs_ref": int(theirs.ran_global_ba.sum()),
              "inactive_edges": int(theirs.pg.ii_inac.numel())}, t0)
        del ours, theirs
    return 0


if __name__ == "__main__":
    sys.exit(main())

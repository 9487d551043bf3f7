"""Lock-step runner: dpvo_amd's tracker and the REFERENCE's own tracker (oracle/ref_pipeline.py: its Python + its native kernels,
compiled for gfx950) on the same frames, weights and random draws; per-frame comparison of the state both leave behind.
Shared by tests/test_zz_ref_pipeline.py and tools/ref_parity.py (which commits the measured distances to profiles/)."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

INT_KEYS = ("ii", "jj", "kk", "ii_inac", "jj_inac", "kk_inac", "tstamps")


def stream(n, ht, wd, device, seed=1234):
    from bench import make_stream
    return make_stream(n, ht, wd, device, seed=seed)


def build_pair(dev, ht=480, wd=640, M=96, seed=1234, feed=True, defer=True, overlap=True, buffer=256, ref_over=None, ours=True,
               delta_scale=1.0, delta_bias=None, **cfg_over):
    """(ours, theirs, cfg): both trackers on default.yaml + overrides, the same random-init VONet weights (strict load on both sides),
    the initialisation probe accepted on both (random weights)"""
    from oracle import ref_pipeline as RP
    from dpvo_amd.config import cfg as base_cfg, DEFAULT_YAML
    from dpvo_amd.dpvo import DPVO
    from dpvo_amd.net import VONet
    cfg = base_cfg.clone()
    cfg.merge_from_dict(DEFAULT_YAML)
    cfg.PATCHES_PER_FRAME = M
    cfg.BUFFER_SIZE = buffer
    for k, v in cfg_over.items():
        cfg[k] = v
    torch.manual_seed(seed)
    net = VONet()
    if delta_scale != 1.0:
        # random-init weights make the flow head emit ~1 px of noise per edge, and a monocular tracker fed noise runs away in scale
        # (extent x 2000 within 40 frames: tools/ref_parity.py scenario A); a smaller head keeps the SAME arithmetic in a bounded regime
        with torch.no_grad():
            net.update.d[1].weight.mul_(delta_scale)
            net.update.d[1].bias.mul_(delta_scale)
    if delta_bias is not None:
        # a COHERENT flow on top: every edge is asked to move by the same (bx, by) pixels per update -- an image-wide shift, i.e. what a
        # rotating camera produces, which bundle adjustment can explain with the poses (well conditioned), where per-edge noise of a
        # static scene can only be absorbed by depths and scale (ill conditioned: the run-away of tools/ref_parity.py scenario A)
        with torch.no_grad():
            net.update.d[1].bias.add_(torch.tensor(delta_bias, dtype=net.update.d[1].bias.dtype))
    sd = {k: v.detach().clone() for k, v in net.state_dict().items()}        # (DPVO casts the encoder towers to f16 in place)
    if ours:
        ours = DPVO(cfg, net, ht=ht, wd=wd, device=dev, defer_keyframe=defer, overlap_encoders=overlap)
        ours.motion_probe = lambda: 1.0e9
    else:       # (reference against reference: the second tracker is another instance of the reference's class)
        ours = RP.make_tracker(RP.make_cfg(cfg), sd, ht, wd, accept_probe=True, feed_encoders=feed)
    theirs = RP.make_tracker(RP.make_cfg(cfg, **(ref_over or {})), sd, ht, wd, accept_probe=True, feed_encoders=feed)
    theirs._flows = []
    real_mm = theirs.motionmag

    def mm(i, j):                       # (records what the reference's flow test saw; the value is passed through untouched)
        v = real_mm(i, j)
        theirs._flows.append(v)
        return v
    theirs.motionmag = mm
    return ours, theirs, cfg


class BACapture:
    """Records every fastba.BA call of the reference's tracker (dpvo.py:323-324,353-354): the arguments as they are when the call
    starts (clones of poses / patches: cuda_ba updates them in place) and the poses / patches it leaves.  One hook on the staged
    reference module serves every tracker; `on` switches the recording."""

    def __init__(self, RP):
        self.calls, self.on = [], False
        fb = RP.load().dpvo_module.fastba
        if not hasattr(fb, "_real_BA"):
            fb._real_BA = fb.BA
        self.real = real = fb._real_BA
        cap = self

        def BA(poses, patches, intrinsics, target, weight, lmbda, ii, jj, kk, t0, t1, M, iterations, eff_impl=False):
            if not cap.on:
                return real(poses, patches, intrinsics, target, weight, lmbda, ii, jj, kk, t0, t1, M=M, iterations=iterations, eff_impl=eff_impl)
            c = dict(poses=poses.data.clone(), patches=patches.clone(), intrinsics=intrinsics.clone(), target=target.clone(),
                     weight=weight.clone(), lmbda=lmbda.clone(), ii=ii.clone(), jj=jj.clone(), kk=kk.clone(), t0=int(t0), t1=int(t1), M=M,
                     iterations=iterations, eff_impl=eff_impl)
            out = real(poses, patches, intrinsics, target, weight, lmbda, ii, jj, kk, t0, t1, M=M, iterations=iterations, eff_impl=eff_impl)
            c["poses_after"], c["patches_after"] = poses.data.clone(), patches.clone()
            cap.calls.append(c)
            return out
        fb.BA = BA

    def rerun(self, c, ba=None, target=None, weight=None):
        """the captured call once more on fresh copies of its inputs -> (poses, patches) it leaves.  ba: another implementation with
        the reference's signature (dpvo_amd.fastba.BA); target / weight: another update operator's outputs for the same edges"""
        p, pt = c["poses"].clone(), c["patches"].clone()
        tg = c["target"] if target is None else target.reshape(c["target"].shape).to(c["target"].dtype)
        wg = c["weight"] if weight is None else weight.reshape(c["weight"].shape).to(c["weight"].dtype)
        (ba or self.real)(p, pt, c["intrinsics"].clone(), tg.clone(), wg.clone(), c["lmbda"].clone(), c["ii"], c["jj"], c["kk"], c["t0"], c["t1"],
                          M=c["M"], iterations=c["iterations"], eff_impl=c["eff_impl"])
        return p, pt


def pose_dist(a, b, n):
    """max |component difference| of the first n poses of two [1,N,7] / [N,7] tensors (quaternion sign fixed)"""
    a, b = a.reshape(-1, 7)[:n].double(), b.reshape(-1, 7)[:n].double()
    sgn = torch.sign((a[:, 3:] * b[:, 3:]).sum(-1, keepdim=True))
    sgn[sgn == 0] = 1
    return float(torch.cat([a[:, :3] - b[:, :3], a[:, 3:] * sgn - b[:, 3:]], -1).abs().max()) if n > 0 else 0.0


def our_outputs_at_ba_time(c, ours, e_inac_before):
    """OUR update operator's targets / weights for exactly the edge list of the captured call `c`.  When the frame is over the
    keyframe step has split that list -- kept edges stay (same order) in the active store, removed ones sit at the tail of the
    inactive store (dpvo.py:305-310; a global BA's list starts with the inactive store as it was, dpvo.py:315-319) -- so the arrays
    are put together again through the (patch, target frame) keys, and every piece is checked against the captured list.
    None when that does not work out (a keyframe was dropped in between: frames and patches were renumbered)."""
    key = lambda kk, jj: kk.long() * 65536 + jj.long()
    pg = ours.pg
    kc = key(c["kk"], c["jj"])
    off = e_inac_before if c["eff_impl"] else 0
    if off > kc.numel() or pg.kk_inac.numel() < off or (off and not torch.equal(kc[:off], key(pg.kk_inac[:off], pg.jj_inac[:off]))):
        return None
    ka, kept = kc[off:], key(pg.kk, pg.jj)
    n_rem = ka.numel() - kept.numel()
    if n_rem < 0 or pg.kk_inac.numel() < e_inac_before + n_rem:
        return None
    sl = slice(e_inac_before, e_inac_before + n_rem)
    mask = torch.isin(ka, kept)
    if int(mask.sum()) != kept.numel() or not torch.equal(ka[mask], kept) or not torch.equal(ka[~mask], key(pg.kk_inac[sl], pg.jj_inac[sl])):
        return None
    out = []
    for act, inac in ((pg.target[0], pg.target_inac[0]), (pg.weight[0], pg.weight_inac[0])):
        v = torch.empty(ka.numel(), 2, dtype=act.dtype, device=act.device)
        v[mask] = act
        v[~mask] = inac[sl]
        out.append(torch.cat((inac[:off], v)) if off else v)
    return out


def exact_ba(c, target=None, weight=None):
    """the captured call solved by the f64 CPU oracle (oracle/dpvo_oracle.c: cuda_ba restated, ba_cuda.cu:232-582) -> poses [N,7] as a
    tensor: what both f32 implementations approximate"""
    import oracle
    n = c["t1"]
    cpu = lambda t: t.detach().double().cpu().numpy()
    tg = c["target"] if target is None else target
    wg = c["weight"] if weight is None else weight
    P = c["patches"].shape[-1]
    poses, _patches, _info, _rt = oracle.ba(cpu(c["poses"]).reshape(-1, 7), cpu(c["patches"]).reshape(-1, 3, P, P), cpu(c["intrinsics"]).reshape(-1, 4),
                                           cpu(tg).reshape(-1, 2), cpu(wg).reshape(-1, 2), float(c["lmbda"].reshape(-1)[0]),
                                           c["ii"].cpu().numpy(), c["jj"].cpu().numpy(), c["kk"].cpu().numpy(), c["t0"], n,
                                           iterations=c["iterations"], dtype=np.float64)
    return torch.from_numpy(poses)


def attribute(cap, c, ours, e_inac_before=0, reruns=3, pose_max=0.0, lim0=float("inf")):
    """One captured BA call of the reference, taken apart:
      yard      -- the reference against ITSELF: the same call re-run `reruns` times on the same inputs (float atomics in
                   ba_cuda.cu:335-373 are its only source of difference): the one-step reference-vs-reference spread on this box
      ba_dist   -- OUR bundle adjustment (dpvo_amd.fastba.BA through the C ABI) on the reference's inputs against the reference's result
      attr_dist -- the reference's OWN bundle adjustment on OUR update operator's targets / weights against OUR poses of this frame:
                   small means the whole pose difference of the frame is what the reference's solver makes of the (separately asserted)
                   differences of the update operator's outputs -- i.e. conditioning, not an implementation
    and (global BA: only on frames where one of the distances exceeds the plain tolerance lim0, the dense f64 solve is slow):
      ref_exact / ours_exact -- the reference's result and ours (same inputs) against the f64 solution of the same two Gauss-Newton
                   steps: how far each f32 implementation is from what both approximate.  Re-running the reference only samples the
                   ORDER noise of its atomics; an implementation with a different summation tree differs from it by more than that
                   wherever the step is ill-conditioned, and the honest question there is which of the two is closer to the exact step
      attr_exact -- the f64 solve on OUR targets / weights against OUR poses of this frame"""
    from dpvo_amd import fastba as our_fastba
    n = c["t1"]
    out = {"yard": 0.0, "eff_impl": bool(c["eff_impl"]), "E_ba": int(c["ii"].numel())}
    # step_ref: how far the reference's own call moved the poses (|poses_after - poses_before|, same metric as every distance below): the
    # scale a one-step tolerance has to be tied to -- a bundle adjustment that did NOTHING is exactly step_ref away from the reference
    out["step_ref"] = pose_dist(c["poses_after"], c["poses"], n)
    for _ in range(reruns):
        p, _pt = cap.rerun(c)
        out["yard"] = max(out["yard"], pose_dist(p, c["poses_after"], n))
    p_ours, pt = cap.rerun(c, ba=our_fastba.BA)
    out["ba_dist"] = pose_dist(p_ours, c["poses_after"], n)
    m = n * c["M"]
    da, db = pt.reshape(-1, 3, 3, 3)[:m, 2, 1, 1], c["patches_after"].reshape(-1, 3, 3, 3)[:m, 2, 1, 1]
    out["ba_depth_rel_p90"] = float(torch.quantile(((da - db).abs() / db.abs().clamp_min(1e-2))[::max(1, m // 4096)], 0.9)) if m else 0.0
    tw = our_outputs_at_ba_time(c, ours, e_inac_before)
    if tw is not None and ours.n == n:
        p, _pt = cap.rerun(c, target=tw[0], weight=tw[1])
        out["attr_dist"] = pose_dist(p, ours.pg.poses_, n)
    else:
        tw = None
    # (a local BA's f64 solve takes ~30 ms at E = 47 712: every frame gets one; a global BA's takes seconds: only where a distance
    #  exceeds the plain tolerance, which is rare by construction)
    if not c["eff_impl"] or (max(out["ba_dist"], pose_max) > lim0 and int(c["ii"].numel()) <= 400000):
        ex = exact_ba(c)
        out["ref_exact"] = pose_dist(c["poses_after"].cpu(), ex, n)
        out["ours_exact"] = pose_dist(p_ours.cpu(), ex, n)
        if tw is not None and pose_max > lim0:
            out["attr_exact"] = pose_dist(exact_ba(c, tw[0], tw[1]), ours.pg.poses_.cpu(), n)
    return out


def compare(so, sr):
    """distances between two snapshots (oracle.ref_pipeline.snapshot) of the same frame"""
    bad = [k for k in ("n", "m", "counter", "delta_keys") if so[k] != sr[k]]
    bad += [k for k in INT_KEYS if so[k].shape != sr[k].shape or not np.array_equal(so[k], sr[k])]
    d = {"int_equal": not bad}
    if bad:
        d["int_mismatch"] = {k: ([int(np.size(so[k])), int(np.size(sr[k]))] if k in INT_KEYS else [so[k], sr[k]]) for k in bad}
    if not d["int_equal"] or so["n"] == 0:
        return d
    n = so["n"]
    d["patch_xy_equal"] = bool(np.array_equal(so["patches"][:, :, :2], sr["patches"][:, :, :2]))
    d["intrinsics_equal"] = bool(np.array_equal(so["intrinsics"], sr["intrinsics"]))
    d["colors_maxdiff"] = int(np.abs(so["colors"].astype(np.int32) - sr["colors"].astype(np.int32)).max())
    Pg, Pr = so["poses"], sr["poses"]
    sgn = np.sign((Pg[:, 3:] * Pr[:, 3:]).sum(-1, keepdims=True)); sgn[sgn == 0] = 1
    d["pose_max"] = float(np.abs(np.concatenate([Pg[:, :3] - Pr[:, :3], Pg[:, 3:] * sgn - Pr[:, 3:]], -1)).max())
    d["trans_rms"] = float(np.sqrt(((Pg[:, :3] - Pr[:, :3]) ** 2).sum(-1).mean()))
    d["extent"] = float(np.linalg.norm(Pr[:, :3].max(0) - Pr[:, :3].min(0)))
    dg, dr = so["patches"][:, :, 2, 1, 1], sr["patches"][:, :, 2, 1, 1]
    rel = np.abs(dg - dr) / np.maximum(np.abs(dr), 1e-2)
    d["depth_rel_p50"], d["depth_rel_p90"], d["depth_rel_max"] = (float(np.quantile(rel, .5)), float(np.quantile(rel, .9)), float(rel.max()))
    d["finite"] = bool(np.isfinite(Pg).all() and np.isfinite(dg).all() and np.isfinite(Pr).all())
    return d


def sync_from_reference(ours, theirs):
    """teacher forcing: our tracker continues from the REFERENCE's float state (poses, patches incl. depths, hidden state).  The
    integer state is never touched -- it has to be identical on its own."""
    n = theirs.n
    ours.pg.poses_[:n].copy_(theirs.pg.poses_[:n])
    ours.pg.patches_[:n].copy_(theirs.pg.patches_[:n])
    ours.pg.net.copy_(theirs.pg.net.float())


def run_lockstep(ours, theirs, frames, n_frames, intr, feed=True, seed0=5000, flush_each=True, log=None, stop_on_mismatch=True,
                 teacher=False, attribute_ba=False):
    """frame t: same seed -> our call (+ flush) -> [our encoder outputs handed to the reference] -> same seed -> reference call ->
    compare.  Returns the list of per-frame distance records (each carries `t` and the two keyframe decisions).
    teacher: after the comparison our tracker's float state is overwritten with the reference's (sync_from_reference), so that every
    frame measures ONE frame's worth of divergence -- the random-weight tracker is a chaotic recurrence (the reference run twice
    drifts apart as fast as the two implementations do: tools/ref_parity.py scenario R), which makes accumulated distances past
    ~30 frames a statement about the dynamics, not about the implementation."""
    from oracle import ref_pipeline as RP
    recs = []
    n_img = frames.shape[0]
    cap = BACapture(RP) if attribute_ba else None
    for t in range(n_frames):
        img = frames[t % n_img]
        n_o, n_r = ours.n, theirs.n
        torch.manual_seed(seed0 + t)
        if not hasattr(ours, "flush"):              # (reference against reference)
            RP.call(ours, float(t), img, intr)
        else:
            with torch.no_grad():
                ours(float(t), img, intr, image_ready=False)
                if flush_each:
                    ours.flush()
        if feed:
            torch.cuda.synchronize()
            RP.feed(theirs, ours._fmap1_cl[(ours.n - 1) % ours.mem], ours._imap_full)
        nf = len(theirs._flows)
        torch.manual_seed(seed0 + t)
        if cap is not None:
            cap.calls.clear()
            cap.on = True
            e_inac_before = int(theirs.pg.ii_inac.numel())
        RP.call(theirs, float(t), img, intr)
        if cap is not None:
            cap.on = False
        if not flush_each:
            continue
        d = compare(RP.snapshot(ours), RP.snapshot(theirs))
        d["t"] = t
        d["E"] = int(theirs.pg.ii.numel())
        # keyframe decisions of this frame (None before initialisation): dropped <=> n did not grow
        d["drop_ours"], d["drop_ref"] = (ours.n == n_o), (theirs.n == n_r)
        fl = theirs._flows[nf:]
        d["flow_ref"] = 0.5 * (fl[0] + fl[1]) if len(fl) == 2 else None
        lk = getattr(ours, "last_keyframe", None)
        if lk is not None and len(fl) == 2:
            s0, c0, s1, c1 = lk[1]
            d["flow_ours"] = 0.5 * ((s0 / c0 if c0 > 0 else float("nan")) + (s1 / c1 if c1 > 0 else float("nan")))
        else:
            d["flow_ours"] = None
        if d["int_equal"] and hasattr(ours, "flush") and d["E"]:
            dn = (ours.pg.net[0] - theirs.pg.net[0].float())
            d["net_max"], d["net_rms"] = float(dn.abs().max()), float(dn.pow(2).mean().sqrt())
            if hasattr(theirs.pg, "target") and theirs.pg.target.shape[1] == d["E"]:      # (set by the first update, dpvo.py:342-343)
                d["target_max"] = float((ours.pg.target[0] - theirs.pg.target[0]).abs().max())
                d["weight_max"] = float((ours.pg.weight[0] - theirs.pg.weight[0]).abs().max())
        if cap is not None and cap.calls and d["int_equal"] and hasattr(ours, "flush"):
            # (the frame's LAST bundle adjustment: the only one, except in the initialisation frame's 12 updates.  With a keyframe
            #  dropped behind it the poses have moved down a slot and the edges were renumbered: the BA-on-the-same-inputs figures do
            #  not care, the attribution against our final poses is not available for that frame)
            with torch.no_grad():
                # (all of the frame's bundle adjustments: 12 in the initialisation frame, dpvo.py:461-465 -- the scale its pose
                #  difference is measured against; one otherwise, where it equals step_ref)
                d["step_frame"] = float(sum(pose_dist(c["poses_after"], c["poses"], c["t1"]) for c in cap.calls))
                d.update(attribute(cap, cap.calls[-1], ours, e_inac_before=e_inac_before, pose_max=d.get("pose_max", 0.0),
                                   lim0=1e-3 * max(1.0, d.get("extent", 0.0))))
            cap.calls.clear()
        recs.append(d)
        if log is not None:
            log(d)
        if stop_on_mismatch and not d["int_equal"]:
            break
        if teacher and d["int_equal"]:
            sync_from_reference(ours, theirs)
    return recs


def summarise(recs):
    ok = [r for r in recs if r.get("int_equal") and "pose_max" in r]
    out = {"frames": len(recs), "int_equal_frames": sum(bool(r.get("int_equal")) for r in recs),
           "first_int_mismatch": next(({"t": r["t"], **r.get("int_mismatch", {})} for r in recs if not r.get("int_equal")), None)}
    if ok:
        out.update(pose_max=max(r["pose_max"] for r in ok), trans_rms_max=max(r["trans_rms"] for r in ok),
                   trans_rms_last=ok[-1]["trans_rms"], extent_last=ok[-1]["extent"],
                   depth_rel_p50=max(r["depth_rel_p50"] for r in ok), depth_rel_p90=max(r["depth_rel_p90"] for r in ok),
                   colors_maxdiff=max(r["colors_maxdiff"] for r in ok), patch_xy_equal=all(r["patch_xy_equal"] for r in ok),
                   intrinsics_equal=all(r["intrinsics_equal"] for r in ok), finite=all(r["finite"] for r in ok),
                   E_last=ok[-1]["E"], net_max=max((r.get("net_max", 0.0) for r in ok), default=None),
                   net_rms=max((r.get("net_rms", 0.0) for r in ok), default=None),
                   target_max=max((r.get("target_max", 0.0) for r in ok), default=None),
                   weight_max=max((r.get("weight_max", 0.0) for r in ok), default=None),
                   pose_max_first28=max((r["pose_max"] for r in ok if r["t"] < 28), default=None),
                   pose_max_first24=max((r["pose_max"] for r in ok if r["t"] < 24), default=None),
                   pose_max_first16=max((r["pose_max"] for r in ok if r["t"] < 16), default=None),
                   above_1e3=[{k: (float(f"{r[k]:.3g}") if isinstance(r.get(k), float) else r.get(k)) for k in
                               ("t", "pose_max", "extent", "yard", "step_ref", "ba_dist", "attr_dist", "ref_exact", "ours_exact", "attr_exact", "eff_impl")}
                              for r in ok if max(r["pose_max"], r.get("ba_dist", 0.0)) > 1e-3 * max(1.0, r["extent"])],
                   pose_series=[[r["t"], float(f"{r['pose_max']:.3g}"), float(f"{r['extent']:.3g}")] for r in ok[::4]])
    at = [r for r in ok if "ba_dist" in r]
    if at:
        out.update(yard_max=max(r["yard"] for r in at), ba_dist_max=max(r["ba_dist"] for r in at),
                   step_ref_min_med_max=[float(f"{v:.3g}") for v in np.quantile([r["step_ref"] for r in at], [0.0, 0.5, 1.0])],
                   step_series=[[r["t"], float(f"{r['step_ref']:.3g}"), float(f"{r['yard']:.3g}"), float(f"{r['ba_dist']:.3g}"),
                                 float(f"{r['pose_max']:.3g}"), float(f"{r.get('ref_exact', float('nan')):.3g}"), float(f"{r['extent']:.3g}")] for r in at],
                   attr_dist_max=max((r["attr_dist"] for r in at if "attr_dist" in r), default=None),
                   ba_series=[[r["t"], float(f"{r['pose_max']:.3g}"), float(f"{r['yard']:.3g}"), float(f"{r['ba_dist']:.3g}"),
                               float(f"{r.get('attr_dist', float('nan')):.3g}")] for r in at[::4]])
    dec = [(r["t"], r["drop_ours"], r["drop_ref"], r["flow_ours"], r["flow_ref"]) for r in recs if r.get("flow_ref") is not None]
    out["decisions"] = len(dec)
    out["drops_ref"] = sum(1 for x in dec if x[2])
    out["first_decision_mismatch"] = next(({"t": x[0], "ours": x[1], "ref": x[2], "flow_ours": x[3], "flow_ref": x[4]} for x in dec if x[1] != x[2]), None)
    fo = [abs(x[3] - x[4]) for x in dec if x[3] is not None and x[3] == x[3] and x[4] == x[4]]
    out["flow_absdiff_max"] = max(fo) if fo else None
    return out


def trajectory_ate(po, pr):
    """RMS translation difference of two full trajectories [T,7] (x y z qx qy qz qw), no alignment (same frame by construction),
    and the same after a Sim3 (Umeyama) alignment as evaluate_euroc.py:117-119 does with evo"""
    a, b = np.asarray(po)[:, :3], np.asarray(pr)[:, :3]
    raw = float(np.sqrt(((a - b) ** 2).sum(-1).mean()))
    from dpvo_amd import traj as T
    try:
        aligned = float(T.ate_rmse(a, b))
    except (AssertionError, np.linalg.LinAlgError):       # degenerate trajectories cannot be aligned
        aligned = None
    return raw, aligned

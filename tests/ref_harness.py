"""Lock-step runner: dpvo_amd's tracker and the REFERENCE's own tracker (oracle/ref_pipeline.py: its Python + its native kernels,
compiled for gfx950) on the same frames, weights and random draws; per-frame comparison of the state both leave behind.
Shared by tests/test_gpu_ref_pipeline.py and tools/ref_parity.py (which commits the measured distances to profiles/)."""
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
               delta_scale=1.0, **cfg_over):
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
                 teacher=False):
    """frame t: same seed -> our call (+ flush) -> [our encoder outputs handed to the reference] -> same seed -> reference call ->
    compare.  Returns the list of per-frame distance records (each carries `t` and the two keyframe decisions).
    teacher: after the comparison our tracker's float state is overwritten with the reference's (sync_from_reference), so that every
    frame measures ONE frame's worth of divergence -- the random-weight tracker is a chaotic recurrence (the reference run twice
    drifts apart as fast as the two implementations do: tools/ref_parity.py scenario R), which makes accumulated distances past
    ~30 frames a statement about the dynamics, not about the implementation."""
    from oracle import ref_pipeline as RP
    recs = []
    n_img = frames.shape[0]
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
        RP.call(theirs, float(t), img, intr)
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
                   pose_series=[[r["t"], float(f"{r['pose_max']:.3g}"), float(f"{r['extent']:.3g}")] for r in ok[::4]])
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

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


def build_pair(dev, ht=480, wd=640, M=96, seed=1234, feed=True, defer=True, overlap=True, buffer=256, **cfg_over):
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
    sd = {k: v.detach().clone() for k, v in net.state_dict().items()}        # (DPVO casts the encoder towers to f16 in place)
    ours = DPVO(cfg, net, ht=ht, wd=wd, device=dev, defer_keyframe=defer, overlap_encoders=overlap)
    ours.motion_probe = lambda: 1.0e9
    theirs = RP.make_tracker(RP.make_cfg(cfg), sd, ht, wd, accept_probe=True, feed_encoders=feed)
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
    d = {"int_equal": all(so[k] == sr[k] for k in ("n", "m", "counter", "delta_keys"))}
    for k in INT_KEYS:
        d["int_equal"] = d["int_equal"] and so[k].shape == sr[k].shape and bool(np.array_equal(so[k], sr[k]))
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


def run_lockstep(ours, theirs, frames, n_frames, intr, feed=True, seed0=5000, flush_each=True, log=None, stop_on_mismatch=True):
    """frame t: same seed -> our call (+ flush) -> [our encoder outputs handed to the reference] -> same seed -> reference call ->
    compare.  Returns the list of per-frame distance records (each carries `t` and the two keyframe decisions)."""
    from oracle import ref_pipeline as RP
    recs = []
    n_img = frames.shape[0]
    for t in range(n_frames):
        img = frames[t % n_img]
        n_o, n_r = ours.n, theirs.n
        torch.manual_seed(seed0 + t)
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
        lk = ours.last_keyframe
        if lk is not None and len(fl) == 2:
            s0, c0, s1, c1 = lk[1]
            d["flow_ours"] = 0.5 * ((s0 / c0 if c0 > 0 else float("nan")) + (s1 / c1 if c1 > 0 else float("nan")))
        else:
            d["flow_ours"] = None
        recs.append(d)
        if log is not None:
            log(d)
        if stop_on_mismatch and not d["int_equal"]:
            break
    return recs


def summarise(recs):
    ok = [r for r in recs if r.get("int_equal") and "pose_max" in r]
    out = {"frames": len(recs), "int_equal_frames": sum(bool(r.get("int_equal")) for r in recs),
           "first_int_mismatch": next((r["t"] for r in recs if not r.get("int_equal")), None)}
    if ok:
        out.update(pose_max=max(r["pose_max"] for r in ok), trans_rms_max=max(r["trans_rms"] for r in ok),
                   trans_rms_last=ok[-1]["trans_rms"], extent_last=ok[-1]["extent"],
                   depth_rel_p50=max(r["depth_rel_p50"] for r in ok), depth_rel_p90=max(r["depth_rel_p90"] for r in ok),
                   colors_maxdiff=max(r["colors_maxdiff"] for r in ok), patch_xy_equal=all(r["patch_xy_equal"] for r in ok),
                   intrinsics_equal=all(r["intrinsics_equal"] for r in ok), finite=all(r["finite"] for r in ok),
                   E_last=ok[-1]["E"], n_last=None)
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

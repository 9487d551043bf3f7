"""Trajectory files and ATE without third-party tooling.

The reference's evaluate_*.py hand `DPVO.terminate()`'s output (poses [N,7] = x y z qx qy qz qw, camera-to-world, and
timestamps) to `evo`: `PoseTrajectory3D`, `file_interface.write_tum_trajectory_file`, and
`main_ape.ape(..., align=True, correct_scale=True)` (evaluate_euroc.py:104-119, evaluate_tartan.py:78-99).  evo is not a
dependency here; these are the same operations in numpy: the TUM text format, timestamp association, Umeyama Sim(3) /
SE(3) alignment and the translation-part RMSE."""
import numpy as np


def save_tum(path, tstamps, poses):
    """one line per pose: `t x y z qx qy qz qw` (what evo's write_tum_trajectory_file emits)"""
    tstamps, poses = np.asarray(tstamps, np.float64).reshape(-1), np.asarray(poses, np.float64).reshape(-1, 7)
    assert tstamps.size == poses.shape[0]
    with open(path, "w") as f:
        for t, p in zip(tstamps, poses):
            f.write(" ".join([repr(float(t))] + [repr(float(v)) for v in p]) + "\n")


def load_tum(path):
    """-> (tstamps [N], poses [N,7]); comment lines (#) and blank lines are skipped, commas tolerated"""
    rows = []
    with open(path) as f:
        for line in f:
            line = line.strip()
            if not line or line.startswith("#"):
                continue
            rows.append([float(v) for v in line.replace(",", " ").split()])
    a = np.asarray(rows, np.float64).reshape(-1, 8)
    return a[:, 0], a[:, 1:]


def associate(t_est, t_ref, max_diff=0.01):
    """greedy nearest-timestamp matching (evo's sync.associate_trajectories): index pairs (i_est, i_ref)"""
    t_est, t_ref = np.asarray(t_est, np.float64), np.asarray(t_ref, np.float64)
    order = np.argsort(t_ref)
    pos = np.searchsorted(t_ref[order], t_est)
    ie, ir, used = [], [], set()
    for i, p in enumerate(pos):
        best, bd = -1, max_diff
        for q in (p - 1, p):
            if 0 <= q < order.size:
                d = abs(t_ref[order[q]] - t_est[i])
                if d <= bd and int(order[q]) not in used:
                    best, bd = int(order[q]), d
        if best >= 0:
            used.add(best); ie.append(i); ir.append(best)
    return np.asarray(ie, np.int64), np.asarray(ir, np.int64)


def umeyama(X, Y, with_scale=True):
    """least-squares similarity (s, R, t) with Y ~ s R X + t (Umeyama 1991); X, Y [N,3]"""
    X, Y = np.asarray(X, np.float64), np.asarray(Y, np.float64)
    mx, my = X.mean(0), Y.mean(0)
    Xc, Yc = X - mx, Y - my
    U, D, Vt = np.linalg.svd(Yc.T @ Xc / X.shape[0])
    S = np.eye(3)
    if np.linalg.det(U) * np.linalg.det(Vt) < 0:
        S[2, 2] = -1.0
    R = U @ S @ Vt
    var = (Xc ** 2).sum() / X.shape[0]
    s = float((D * np.diag(S)).sum() / var) if (with_scale and var > 0) else 1.0      # (a trajectory that never moved: no scale)
    return s, R, my - s * R @ mx


def ate_rmse(est_xyz, ref_xyz, align=True, correct_scale=True):
    """translation-part absolute trajectory error (evo APE, PoseRelation.translation_part, statistics 'rmse')"""
    est_xyz, ref_xyz = np.asarray(est_xyz, np.float64).reshape(-1, 3), np.asarray(ref_xyz, np.float64).reshape(-1, 3)
    assert est_xyz.shape == ref_xyz.shape and est_xyz.shape[0] >= 3
    if align:
        s, R, t = umeyama(est_xyz, ref_xyz, with_scale=correct_scale)
        est_xyz = (s * (R @ est_xyz.T)).T + t
    return float(np.sqrt(((est_xyz - ref_xyz) ** 2).sum(-1).mean()))


def ate_from_files(est_path, ref_path, max_diff=0.01, correct_scale=True):
    te, pe = load_tum(est_path)
    tr, pr = load_tum(ref_path)
    ie, ir = associate(te, tr, max_diff)
    return ate_rmse(pe[ie, :3], pr[ir, :3], align=True, correct_scale=correct_scale)

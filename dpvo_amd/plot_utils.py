"""Result writers with the names the reference's scripts import from dpvo/plot_utils.py (`plot_trajectory`, `save_ply`,
`save_output_for_COLMAP`).  No evo / plyfile dependency: trajectories are anything with `positions_xyz`, `orientations_quat_wxyz`
and `timestamps` (evo's PoseTrajectory3D has them) or a plain (timestamps, poses[N,7] = x y z qx qy qz qw) pair; alignment uses
dpvo_amd.traj (Umeyama); PLY / COLMAP text files are written directly.  matplotlib is imported only when a plot is requested."""
from pathlib import Path

import numpy as np

from . import traj as T


def _xyz_t(tr):
    if hasattr(tr, "positions_xyz"):
        return np.asarray(tr.positions_xyz, np.float64), np.asarray(tr.timestamps, np.float64)
    ts, poses = tr
    return np.asarray(poses, np.float64)[:, :3], np.asarray(ts, np.float64)


def plot_trajectory(pred_traj, gt_traj=None, title="", filename="", align=True, correct_scale=True):
    """x-z plot of the (optionally Sim(3)-aligned) estimate over the ground truth"""
    import matplotlib
    matplotlib.use("Agg")
    import matplotlib.pyplot as plt
    pe, te = _xyz_t(pred_traj)
    fig = plt.figure(figsize=(8, 8))
    ax = fig.add_subplot(111)
    ax.set_title(title); ax.set_xlabel("x [m]"); ax.set_ylabel("z [m]"); ax.set_aspect("equal", adjustable="datalim")
    if gt_traj is not None:
        pr, tr = _xyz_t(gt_traj)
        ie, ir = T.associate(te, tr)
        if align and ie.size >= 3:
            s, R, t = T.umeyama(pe[ie], pr[ir], with_scale=correct_scale)
            pe = (s * (R @ pe.T)).T + t
        ax.plot(pr[:, 0], pr[:, 2], "--", color="gray", label="Ground Truth")
    ax.plot(pe[:, 0], pe[:, 2], "-", color="blue", label="Predicted")
    ax.legend()
    if filename:
        Path(filename).parent.mkdir(parents=True, exist_ok=True)
        fig.savefig(filename)
        print(f"Saved {filename}")
    plt.close(fig)


def save_ply(name, points, colors):
    """binary little-endian PLY point cloud `<name>.ply` (x y z float32, red green blue uint8)"""
    points = np.asarray(points, np.float32).reshape(-1, 3)
    colors = np.asarray(colors, np.uint8).reshape(-1, 3)
    rec = np.empty(points.shape[0], dtype=[("x", "<f4"), ("y", "<f4"), ("z", "<f4"), ("red", "u1"), ("green", "u1"), ("blue", "u1")])
    rec["x"], rec["y"], rec["z"] = points.T
    rec["red"], rec["green"], rec["blue"] = colors.T
    path = f"{name}.ply"
    with open(path, "wb") as f:
        f.write(("ply\nformat binary_little_endian 1.0\nelement vertex %d\nproperty float x\nproperty float y\nproperty float z\n"
                 "property uchar red\nproperty uchar green\nproperty uchar blue\nend_header\n" % points.shape[0]).encode())
        f.write(rec.tobytes())
    print(f"Saved {path}")


def save_output_for_COLMAP(name, traj, points, colors, fx, fy, cx, cy, H=480, W=640):
    """COLMAP text model (cameras.txt, images.txt, points3D.txt) of the sparse map, world scaled by 10 like the reference"""
    out = Path(name)
    out.mkdir(parents=True, exist_ok=True)
    scale = 10.0
    if hasattr(traj, "positions_xyz"):
        xyz = np.asarray(traj.positions_xyz, np.float64)
        qwxyz = np.asarray(traj.orientations_quat_wxyz, np.float64)
    else:
        _, poses = traj
        poses = np.asarray(poses, np.float64)
        xyz, qwxyz = poses[:, :3], poses[:, [6, 3, 4, 5]]
    (out / "cameras.txt").write_text(f"1 PINHOLE {W} {H} {fx} {fy} {cx} {cy}\n")
    lines = []
    for i, (p, q) in enumerate(zip(xyz, qwxyz)):
        # COLMAP stores world-to-camera: invert the camera-to-world pose (q, p)
        w, x, y, z = q / np.linalg.norm(q)
        R = np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                      [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                      [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])
        t = -R.T @ (p * scale)
        lines.append(f"{i + 1} {w} {-x} {-y} {-z} {t[0]} {t[1]} {t[2]} 1 image_{i:05d}.png\n\n")
    (out / "images.txt").write_text("".join(lines))
    pts = np.asarray(points, np.float64).reshape(-1, 3) * scale
    col = np.asarray(colors, np.uint8).reshape(-1, 3)
    with open(out / "points3D.txt", "w") as f:
        for i, (p, c) in enumerate(zip(pts, col)):
            f.write(f"{i + 1} {p[0]} {p[1]} {p[2]} {c[0]} {c[1]} {c[2]} 0.0\n")
    print(f"Saved COLMAP-compatible reconstruction in {out.resolve()}")

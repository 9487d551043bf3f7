"""dpvo_amd/traj.py: TUM files, timestamp association, Umeyama alignment and ATE (evo's operations, evaluate_euroc.py:104-119)."""
import numpy as np

from dpvo_amd import traj


def _rot(rng):
    q = rng.standard_normal(4); q /= np.linalg.norm(q)
    x, y, z, w = q
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                     [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                     [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])


def test_umeyama_recovers_similarity_and_ate_is_zero():
    rng = np.random.default_rng(0)
    X = np.cumsum(rng.standard_normal((200, 3)) * 0.05, 0)
    R, t, s = _rot(rng), rng.standard_normal(3), 2.7
    Y = (s * (R @ X.T)).T + t
    s2, R2, t2 = traj.umeyama(X, Y)
    assert abs(s2 - s) < 1e-9 and np.abs(R2 - R).max() < 1e-9 and np.abs(t2 - t).max() < 1e-9
    assert traj.ate_rmse(X, Y) < 1e-9
    assert traj.ate_rmse(X, Y, correct_scale=False) > 1e-2            # SE(3) alignment cannot absorb the scale
    noise = rng.standard_normal(X.shape) * 0.01
    assert abs(traj.ate_rmse(X + noise, Y / 1.0) - 0.01 * s * np.sqrt(3)) < 0.01 * s      # ~ sigma * s * sqrt(3)


def test_tum_round_trip_and_association(tmp_path):
    rng = np.random.default_rng(1)
    t = np.arange(50) * 0.05 + 1403636579.7
    p = np.concatenate([rng.standard_normal((50, 3)), rng.standard_normal((50, 4))], 1)
    p[:, 3:] /= np.linalg.norm(p[:, 3:], axis=1, keepdims=True)
    f = tmp_path / "est.txt"
    traj.save_tum(str(f), t, p)
    t2, p2 = traj.load_tum(str(f))
    assert np.array_equal(t, t2) and np.array_equal(p, p2)            # repr() round-trips doubles exactly
    g = tmp_path / "ref.txt"
    traj.save_tum(str(g), t[::2] + 0.004, p[::2])                     # half rate, 4 ms clock offset
    ie, ir = traj.associate(t, t[::2] + 0.004, max_diff=0.01)
    assert np.array_equal(ie, np.arange(0, 50, 2)) and np.array_equal(ir, np.arange(25))
    assert traj.ate_from_files(str(f), str(g)) < 1e-9

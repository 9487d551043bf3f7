"""dpvo_solve_system (dpvo_amd/csrc/pgo.hip) == cuda_ba.solve_system (reference dpvo/fastba/ba.cpp:102-180): one Levenberg-Marquardt step
of the Sim(3) pose-graph optimisation, assembled and solved in f64 on the device, against oracle/pgo_ref.py (the reference's lines restated
in numpy f64: triplet assembly, J^T J, the two damping lines, the solve over the leading block).  Stated tolerance: the result is f32 (the
reference casts its double solution to float, :154), so |HIP - oracle| <= 2e-6 x max(1, |oracle|) per component."""
import numpy as np
import pytest
import torch

from dpvo_amd import fastba
from dpvo_amd import _lib as L

pytestmark = pytest.mark.gpu


def _graph(n, n_loop, seed, dup=False):
    """a pose graph as the classical loop closure builds it: the odometry chain i -> i + 1 plus a few long-range loop edges; Jacobian
    blocks near -Ad / +I like a relative-pose residual's, residuals O(0.1)"""
    rng = np.random.default_rng(seed)
    ii = np.arange(n - 1); jj = ii + 1
    li = rng.integers(0, n - 10, n_loop); lj = li + rng.integers(5, 10 + (n - 10 - li) // 2 + 1)
    ii = np.concatenate([ii, li]); jj = np.concatenate([jj, np.minimum(lj, n - 1)])
    if dup:                                               # the same edge twice, and an edge in both directions: triplets add up
        ii = np.concatenate([ii, ii[:3], jj[3:6]]); jj = np.concatenate([jj, jj[:3], ii[3:6]])
    r = len(ii)
    Ji = (-np.eye(7)[None] + 0.2 * rng.standard_normal((r, 7, 7))).astype(np.float32)
    Jj = (np.eye(7)[None] + 0.2 * rng.standard_normal((r, 7, 7))).astype(np.float32)
    res = (0.1 * rng.standard_normal((r, 7))).astype(np.float32)
    return Ji, Jj, ii.astype(np.int64), jj.astype(np.int64), res


def _run(dev, Ji, Jj, ii, jj, res, ep, lm, freen):
    d = lambda a: torch.from_numpy(a).to(dev)
    out, = fastba.solve_system(d(Ji), d(Jj), d(ii), d(jj), d(res), ep, lm, freen)
    torch.cuda.synchronize()
    return out.cpu().numpy()


@pytest.mark.parametrize("n,n_loop,ep,lm,freen,dup", [
    (12, 2, 0.0, 1e-6, -1, False),          # perform_updates' defaults (optim_utils.py:214: ep = 0, lmbda = 1e-6), 84 unknowns: three blocks
    (64, 5, 0.0, 1e-6, -1, True),           # 448 = 14 blocks exactly: the bordered row opens a block of its own; duplicate edges
    (101, 8, 1e-3, 1e-4, -1, False),        # ep > 0; 707 unknowns: the bordered row inside the last diagonal block
    (101, 8, 0.0, 1e-6, 60, False),         # fix_opt_window: only the first freen nodes are solved (ba.cpp:102-118)
    (400, 30, 0.0, 1e-6, -1, False),        # 2 800 unknowns, 88 panels
])
def test_solve_system_vs_oracle(oracle, dev, n, n_loop, ep, lm, freen, dup):
    from oracle import pgo_ref
    Ji, Jj, ii, jj, res = _graph(n, n_loop, seed=n + n_loop, dup=dup)
    want = pgo_ref.solve_system(Ji, Jj, ii, jj, res, ep, lm, freen)
    got = _run(dev, Ji, Jj, ii, jj, res, ep, lm, freen)
    nn = int(max(ii.max(), jj.max())) + 1
    assert got.shape == want.shape == (nn, 7) and np.isfinite(got).all()
    err = np.abs(got - want).max() / max(1.0, np.abs(want).max())
    print(f"solve_system n = {nn}, r = {len(ii)}, freen = {freen}: |delta| max {np.abs(want).max():.3g}, |HIP - oracle| / max(1, |oracle|) = {err:.2e}")
    assert err <= 2e-6
    if freen >= 0:
        assert np.all(got[freen:] == 0)
    # ... and the answer solves the damped normal equations of the triplet assembly (ba.cpp:120-153), independently of the oracle's solve
    J, _ = pgo_ref.jacobian_dense(Ji, Jj, ii, jj)
    m = 7 * (freen if freen >= 0 else nn)
    A = (J.T @ J)[:m, :m]
    dg = np.diag(A).copy()
    A[np.diag_indices_from(A)] = dg + dg * np.float64(np.float32(lm)) + np.float64(np.float32(ep))
    b = -(J.T @ res.reshape(-1).astype(np.float64))[:m]
    resid = np.abs(A @ got.reshape(-1)[:m].astype(np.float64) - b).max() / max(1.0, np.abs(b).max())
    assert resid <= 2e-5, resid


def test_solve_system_is_repeatable_to_rounding_and_reports_bad_input(dev):
    Ji, Jj, ii, jj, res = _graph(80, 6, seed=3)
    a = _run(dev, Ji, Jj, ii, jj, res, 0.0, 1e-6, -1)
    b = _run(dev, Ji, Jj, ii, jj, res, 0.0, 1e-6, -1)
    assert np.abs(a - b).max() <= 1e-6 * max(1.0, np.abs(a).max())       # (f64 atomics in the assembly: order noise far below the f32 output)
    jj2 = jj.copy(); jj2[5] = ii[5]                                        # an edge from a node to itself: the reference exits the process
    with pytest.raises(L.DPVOHipError):
        _run(dev, Ji, Jj, ii, jj2, res, 0.0, 1e-6, -1)
    with pytest.raises(L.DPVOHipError):                                    # no damping, zero Jacobians: not positive definite
        _run(dev, 0 * Ji, 0 * Jj, ii, jj, res, 0.0, 0.0, -1)


def test_solve_system_through_the_integration_stub(dev):
    """the ctypes stand-in of INTEGRATION.md (cuda_ba.solve_system) returns what the package's own wrapper returns"""
    import dpvo_amd.integration_stubs as S
    Ji, Jj, ii, jj, res = _graph(40, 3, seed=9)
    d = lambda a: torch.from_numpy(a).to(dev)
    a, = S.cuda_ba.solve_system(d(Ji), d(Jj), d(ii), d(jj), d(res), 0.0, 1e-6, -1)
    b = _run(dev, Ji, Jj, ii, jj, res, 0.0, 1e-6, -1)
    assert np.abs(a.cpu().numpy() - b).max() <= 1e-6 * max(1.0, np.abs(b).max())

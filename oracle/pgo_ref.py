"""TEST INFRASTRUCTURE ONLY -- cuda_ba.solve_system restated in numpy f64 (reference dpvo/fastba/ba.cpp:102-180): the checker of
dpvo_solve_system (dpvo_amd/csrc/pgo.hip).  Follows the reference line by line: the sparse Jacobian from triplets (:120-146), A = J^T J,
b = -J^T v in double (:148-150), the two damping lines (:152-153), the solve over the leading block (:102-118), the cast to float (:154).
A dense solve stands where Eigen's SimplicialCholesky stands: the same linear system, a different factorisation (parity of the native
path is therefore pinned by the normal equations themselves, tests/test_oracle.py, not by a reference binary: Eigen is not in the image)."""
import numpy as np


def jacobian_dense(J_Ginv_i, J_Ginv_j, ii, jj, n=None):
    """J [7 r, 7 n]: rows 7 x + k, columns 7 i + l <- J_Ginv_i[x, k, l], 7 j + l <- J_Ginv_j[x, k, l]; duplicate triplets ADD (setFromTriplets)"""
    Ji = np.asarray(J_Ginv_i, dtype=np.float32).astype(np.float64)
    Jj = np.asarray(J_Ginv_j, dtype=np.float32).astype(np.float64)
    ii, jj = np.asarray(ii, dtype=np.int64), np.asarray(jj, dtype=np.int64)
    r = ii.shape[0]
    if (ii == jj).any():
        raise ValueError("an edge connects a node with itself (ba.cpp:139-140: exit(1))")
    n = int(max(ii.max(), jj.max())) + 1 if n is None else n
    J = np.zeros((7 * r, 7 * n), dtype=np.float64)
    for x in range(r):
        J[7 * x:7 * x + 7, 7 * ii[x]:7 * ii[x] + 7] += Ji[x]
        J[7 * x:7 * x + 7, 7 * jj[x]:7 * jj[x] + 7] += Jj[x]
    return J, n


def solve_system(J_Ginv_i, J_Ginv_j, ii, jj, res, ep, lm, freen):
    """-> delta [n, 7] float32"""
    J, n = jacobian_dense(J_Ginv_i, J_Ginv_j, ii, jj)
    v = np.asarray(res, dtype=np.float32).astype(np.float64).reshape(-1)
    b = -(J.T @ v)
    A = J.T @ J
    d = np.diag(A).copy()
    d = d + d * np.float64(np.float32(lm))            # A.diagonal() += (A.diagonal() * lm);   (lm, ep are floats promoted to double)
    d = d + np.float64(np.float32(ep))                # A.diagonal().array() += ep;
    A[np.diag_indices_from(A)] = d
    m = 7 * n if freen < 0 else min(7 * freen, 7 * n)
    delta = np.zeros(7 * n, dtype=np.float64)
    delta[:m] = np.linalg.solve(A[:m, :m], b[:m])
    return delta.astype(np.float32).reshape(n, 7)

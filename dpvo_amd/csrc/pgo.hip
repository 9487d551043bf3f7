// pgo.hip -- cuda_ba.solve_system (reference dpvo/fastba/ba.cpp:102-180, exported at :188): ONE Levenberg-Marquardt step of the
// Sim(3) pose-graph optimisation behind the reference's classical loop closure (caller: dpvo/loop_closure/optim_utils.py:229,
// perform_updates).  SURVEY 8(f) rank 4, the numerics only (the retrieval / matching pipeline around it and DPViewer stay out of scope).
//
// What the reference does, on the CPU with Eigen in DOUBLE: J [7 r, 7 n] sparse from the per-edge 7x7 blocks J_Ginv_i / J_Ginv_j (edge x
// couples nodes ii[x], jj[x]; :120-146), A = J^T J, b = -J^T res (:148-150), A.diagonal() += A.diagonal() * lm; A.diagonal() += ep
// (:152-153), delta = SimplicialCholesky(A).solve(b) -- over the leading 7 freen x 7 freen block only when freen >= 0, zeros behind it
// (:102-118) -- returned as float [n, 7] (:156-158).
//
// Here: the same arithmetic in f64 on the device, dense.  A pose graph is a chain plus a few loop edges, so A is block-banded and a sparse
// factorisation would do less work; but the solve runs a handful of times per loop closure in a side process of the reference, n is the
// number of keyframes (hundreds to a few thousand), and a dense f64 blocked Cholesky of a 7 000 x 7 000 matrix is ~0.1 s of f64 vector work:
// correctness against the f64 oracle is what matters, so the structure is the simplest one that is exact.
//   pgo_assemble_kernel   one workgroup per edge: the edge's four 7x7 blocks of the LOWER triangle of A and its two slices of b (f64
//                         products and sums of the f32 inputs, f64 atomics: duplicates and shared nodes add up in any order -- a 1e-16
//                         relative effect on a result that is rounded to f32)
//   pgo_damp_kernel       the two damping lines; identity on the padding diagonal
//   pgo_panel_kernel      right-looking blocked Cholesky, 32 x 32 blocks: every workgroup of panel k factorises the diagonal block (cheaper
//                         than a launch and a round trip, as in chol.hip) and solves its own block row against it
//   pgo_update_kernel     trailing update A_ij -= L_ik L_jk^T, one workgroup per tile of the lower triangle
//   b travels as an extra ROW of the matrix (row m): factorising the bordered matrix leaves y = L^-1 b there -- the forward
//   substitution is free (chol.hip does the same);
//   pgo_back_kernel       x = L^-T y, one workgroup, block column by block column from the last.
#include "common.h"

namespace {
namespace pgo {

constexpr int NB = 32;                 // block size
constexpr int S = 7;                   // Sim(3) parameters per node

struct Mat { double* a; int64_t ld; };          // padded work matrix [mp, mp], row-major, lower triangle + the bordered row m
__device__ __forceinline__ double& at(const Mat& M, int64_t r, int64_t c) { return M.a[r * M.ld + c]; }

// edge x: rows / columns 7 i .. 7 i + 6 and 7 j .. 7 j + 6; entries with a row or column >= m (outside the free block) are skipped
__global__ __launch_bounds__(256) void pgo_assemble_kernel(const float* __restrict__ Ji, const float* __restrict__ Jj,
                                                           const int64_t* __restrict__ ii, const int64_t* __restrict__ jj,
                                                           const float* __restrict__ res, int64_t r, Mat M, int64_t m, int* __restrict__ err) {
  __shared__ double J[2][S][S];        // [i / j][residual component k][parameter l]
  __shared__ double v[S];
  const int64_t x = blockIdx.x;
  if (x >= r) return;
  const int t = threadIdx.x;
  const int64_t ni = ii[x], nj = jj[x];
  if (t < 2 * S * S) {
    const int side = t / (S * S), e = t - side * S * S;
    (&J[side][0][0])[e] = (double)(side ? Jj : Ji)[x * S * S + e];
  }
  if (t < S) v[t] = (double)res[x * S + t];
  __syncthreads();
  if (ni == nj || ni < 0 || nj < 0) {                       // ba.cpp:139-140 exits the process here; this library never does
    if (t == 0) atomicExch(err, 1);
    return;
  }
  // 4 blocks x 49 entries = 196 threads; block (a, c): rows of node a, columns of node c, A_ac[l][l'] = sum_k J_a[k][l] J_c[k][l']
  if (t < 4 * S * S) {
    const int blk = t / (S * S), e = t - blk * S * S, l = e / S, lp = e - l * S;
    const int a = blk >> 1, c = blk & 1;
    const int64_t row = (a ? nj : ni) * S + l, col = (c ? nj : ni) * S + lp;
    if (row >= col && row < m) {                            // lower triangle (with the diagonal), inside the free block
      double s = 0.0;
#pragma unroll
      for (int k = 0; k < S; ++k) s += J[a][k][l] * J[c][k][lp];
      unsafeAtomicAdd(&at(M, row, col), s);
    }
  } else if (t < 4 * S * S + 2 * S) {
    const int q = t - 4 * S * S, a = q / S, l = q - a * S;
    const int64_t col = (a ? nj : ni) * S + l;
    if (col < m) {
      double s = 0.0;
#pragma unroll
      for (int k = 0; k < S; ++k) s += J[a][k][l] * v[k];
      unsafeAtomicAdd(&at(M, m, col), -s);                  // b = -J^T res, in the bordered row
    }
  }
}

__global__ void pgo_damp_kernel(Mat M, int64_t m, int64_t mp, double ep, double lm) {
  const int64_t c = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (c >= mp) return;
  if (c < m) {
    double d = at(M, c, c);
    d += d * lm;                                            // A.diagonal() += (A.diagonal() * lm);   ba.cpp:152
    d += ep;                                                // A.diagonal().array() += ep;            ba.cpp:153
    at(M, c, c) = d;
  } else if (c == m) {
    at(M, c, c) = 1.0e300;                                  // the bordered corner: any value above |y|^2 keeps the last pivot real; never read
  } else {
    at(M, c, c) = 1.0;                                      // padding: identity
  }
}

// the 32 x 32 diagonal block in LDS -> its Cholesky factor (lower), in place; 256 threads
__device__ void factor32(double (&D)[NB][NB + 1], int t, int* bad) {
  for (int j = 0; j < NB; ++j) {
    if (t == 0) {
      const double d = D[j][j];
      if (!(d > 0.0)) *bad = 1;
      D[j][j] = sqrt(d);
    }
    __syncthreads();
    if (t > j && t < NB) D[t][j] /= D[j][j];
    __syncthreads();
    // trailing update of the block: rows i > j, columns j < c <= i
    for (int e = t; e < NB * NB; e += 256) {
      const int i = e / NB, c = e - i * NB;
      if (c > j && i >= c) D[i][c] -= D[i][j] * D[c][j];
    }
    __syncthreads();
  }
}

// panel k: workgroup g factorises the diagonal block (all of them do, redundantly, from the UNFACTORED block in M); g == 0 stores the
// factor in Dg[k] -- not in M, which the other workgroups of this launch are still reading -- g > 0 solves block row k + g against it:
// L_ik = A_ik L_kk^-T  (row i of the block: forward substitution along the columns)
__global__ __launch_bounds__(256) void pgo_panel_kernel(Mat M, double* __restrict__ Dg, int64_t k, int* __restrict__ err) {
  __shared__ double D[NB][NB + 1];
  __shared__ double B[NB][NB + 1];
  __shared__ int bad;
  const int t = threadIdx.x;
  const int64_t r0 = k * NB;
  if (t == 0) bad = 0;
  for (int e = t; e < NB * NB; e += 256) {
    const int i = e / NB, c = e - i * NB;
    D[i][c] = c <= i ? at(M, r0 + i, r0 + c) : 0.0;
  }
  __syncthreads();
  factor32(D, t, &bad);
  if (blockIdx.x == 0) {
    for (int e = t; e < NB * NB; e += 256) {
      const int i = e / NB, c = e - i * NB;
      Dg[(k * NB + i) * NB + c] = c <= i ? D[i][c] : 0.0;
    }
    if (t == 0 && bad) atomicExch(err, 2);
    return;
  }
  const int64_t i0 = (k + blockIdx.x) * NB;
  for (int e = t; e < NB * NB; e += 256) {
    const int i = e / NB, c = e - i * NB;
    B[i][c] = at(M, i0 + i, r0 + c);
  }
  __syncthreads();
  // thread i < 32 owns row i of the block: x_c = (b_c - sum_{c' < c} x_c' L[c][c']) / L[c][c]
  if (t < NB) {
    for (int c = 0; c < NB; ++c) {
      double s = B[t][c];
      for (int cp = 0; cp < c; ++cp) s -= B[t][cp] * D[c][cp];
      B[t][c] = s / D[c][c];
    }
  }
  __syncthreads();
  for (int e = t; e < NB * NB; e += 256) {
    const int i = e / NB, c = e - i * NB;
    at(M, i0 + i, r0 + c) = B[i][c];
  }
}

// trailing update with panel k: tile (i, j), k < j <= i:  A_ij -= L_ik L_jk^T.  blockIdx.x enumerates the lower-triangular tiles.
__global__ __launch_bounds__(256) void pgo_update_kernel(Mat M, int64_t k, int64_t nt) {
  __shared__ double Li[NB][NB + 1];
  __shared__ double Lj[NB][NB + 1];
  // tile number -> (a, c) with c <= a < nt:  a (a + 1) / 2 + c
  const int64_t q = blockIdx.x;
  int64_t a = (int64_t)((sqrt(8.0 * (double)q + 1.0) - 1.0) * 0.5);
  while (a * (a + 1) / 2 > q) --a;
  while ((a + 1) * (a + 2) / 2 <= q) ++a;
  const int64_t c = q - a * (a + 1) / 2;
  if (a >= nt) return;
  const int64_t i0 = (k + 1 + a) * NB, j0 = (k + 1 + c) * NB, r0 = k * NB;
  const int t = threadIdx.x;
  for (int e = t; e < NB * NB; e += 256) {
    const int i = e / NB, cc = e - i * NB;
    Li[i][cc] = at(M, i0 + i, r0 + cc);
    Lj[i][cc] = at(M, j0 + i, r0 + cc);
  }
  __syncthreads();
  for (int e = t; e < NB * NB; e += 256) {
    const int i = e / NB, cc = e - i * NB;
    if (a == c && cc > i) continue;                         // diagonal tile: lower part only
    double s = 0.0;
#pragma unroll 8
    for (int p = 0; p < NB; ++p) s += Li[i][p] * Lj[cc][p];
    at(M, i0 + i, j0 + cc) -= s;
  }
}

// x = L^-T y (y = row m of the factorised matrix), block column by block column from the last; ONE workgroup of 1024 threads.
// y is updated in place: after x_k is known, y_j -= L_kj^T x_k for every block j < k.
__global__ __launch_bounds__(1024) void pgo_back_kernel(Mat M, const double* __restrict__ Dg, int64_t m, int64_t nbk, float* __restrict__ delta,
                                                        int64_t n_out) {
  __shared__ double xk[NB];
  __shared__ double D[NB][NB + 1];
  const int t = threadIdx.x;
  double* y = &at(M, m, 0);
  for (int64_t k = nbk - 1; k >= 0; --k) {
    const int64_t r0 = k * NB;
    for (int e = t; e < NB * NB; e += 1024) {
      const int i = e / NB, c = e - i * NB;
      D[i][c] = Dg[(k * NB + i) * NB + c];
    }
    __syncthreads();
    // y_k: from the bordered row in M -- unless that row lies in THIS diagonal block (m not a multiple of the block size): then the
    // panel step left it in the block's factor, row m - r0
    if (t < NB) xk[t] = r0 + t < m ? (m / NB == k ? D[m - r0][t] : y[r0 + t]) : 0.0;
    __syncthreads();
    if (t == 0) {                                           // L_kk^T x = y_k: backward substitution inside the block (32 x 32 / 2 operations)
      for (int c = NB - 1; c >= 0; --c) {
        double s = xk[c];
        for (int i = c + 1; i < NB; ++i) s -= D[i][c] * xk[i];
        xk[c] = s / D[c][c];
      }
    }
    __syncthreads();
    // y_j -= sum_i L[r0 + i][j] x_k[i] for every column j < r0  (the bordered row itself, when it lies in this block, has x = 0:
    // its "L" entries are y and must not be read while y is being updated)
    for (int64_t j = t; j < r0; j += 1024) {
      double s = 0.0;
#pragma unroll 8
      for (int i = 0; i < NB; ++i)
        if (r0 + i < m) s += at(M, r0 + i, j) * xk[i];
      y[j] -= s;
    }
    if (t < NB && r0 + t < m) y[r0 + t] = xk[t];
    __syncthreads();
  }
  for (int64_t c = t; c < n_out; c += 1024) delta[c] = c < m ? (float)y[c] : 0.f;
}

inline int64_t padded(int64_t m) { return ((m + 1 + NB - 1) / NB) * NB; }      // m columns + the bordered row, rounded up to blocks

}  // namespace pgo
}  // namespace

extern "C" size_t dpvo_solve_system_workspace_bytes(int64_t n_nodes, int64_t freen) {
  if (n_nodes <= 0) return 0;
  const int64_t m = pgo::S * (freen >= 0 && freen < n_nodes ? freen : n_nodes);
  const int64_t mp = pgo::padded(m);
  return (size_t)mp * (size_t)mp * sizeof(double) + (size_t)mp * pgo::NB * sizeof(double) + 256;      // work matrix + diagonal factors + status
}

extern "C" int dpvo_solve_system(const float* J_Ginv_i, const float* J_Ginv_j, const int64_t* ii, const int64_t* jj, const float* res,
                                 int64_t r, int64_t n_nodes, float ep, float lm, int64_t freen, float* delta, int32_t* info, void* ws,
                                 size_t ws_bytes, void* stream) {
  using namespace pgo;
  if (r < 0 || n_nodes <= 0 || !delta) return DPVO_E_INVALID;
  if (r > 0 && (!J_Ginv_i || !J_Ginv_j || !ii || !jj || !res)) return DPVO_E_INVALID;
  if (!ws || ws_bytes < dpvo_solve_system_workspace_bytes(n_nodes, freen)) return DPVO_E_WORKSPACE;
  hipStream_t st = (hipStream_t)stream;
  const int64_t nf = freen >= 0 && freen < n_nodes ? freen : n_nodes;       // nodes in the solved block (ba.cpp:102-118)
  const int64_t m = S * nf, mp = padded(m), nbk = mp / NB;
  Mat M{reinterpret_cast<double*>(ws), mp};
  double* Dg = M.a + (size_t)mp * mp;                                        // [nbk][NB][NB] factored diagonal blocks
  int* err = reinterpret_cast<int*>(Dg + (size_t)mp * NB);
  if (hipMemsetAsync(ws, 0, dpvo_solve_system_workspace_bytes(n_nodes, freen), st) != hipSuccess) return DPVO_E_INVALID;
  if (r > 0) hipLaunchKernelGGL(pgo_assemble_kernel, dim3((unsigned)r), dim3(256), 0, st, J_Ginv_i, J_Ginv_j, ii, jj, res, r, M, m, err);
  hipLaunchKernelGGL(pgo_damp_kernel, dim3((unsigned)cdiv64(mp, 256)), dim3(256), 0, st, M, m, mp, (double)ep, (double)lm);
  for (int64_t k = 0; k < nbk; ++k) {
    const int64_t below = nbk - 1 - k;
    hipLaunchKernelGGL(pgo_panel_kernel, dim3((unsigned)(1 + below)), dim3(256), 0, st, M, Dg, k, err);
    if (below > 0)
      hipLaunchKernelGGL(pgo_update_kernel, dim3((unsigned)(below * (below + 1) / 2)), dim3(256), 0, st, M, k, below);
  }
  hipLaunchKernelGGL(pgo_back_kernel, dim3(1), dim3(1024), 0, st, M, Dg, m, (m + NB - 1) / NB, delta, n_nodes * S);
  if (info && hipMemcpyAsync(info, err, sizeof(int), hipMemcpyDeviceToDevice, st) != hipSuccess) return DPVO_E_INVALID;
  DPVO_LAUNCH_CHECK();
  return DPVO_OK;
}

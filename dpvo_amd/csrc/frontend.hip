// frontend.hip -- small fused kernels of the per-frame front-end (the callers either side of the hot path:
// SURVEY.md section 8f "next #2", dpvo/dpvo.py:377-473).  Each replaces a chain of 4..25 tiny torch launches; the
// frame is launch-bound at > 200 fps, so every launch removed is ~8 us of wall time.
#include "common.h"

namespace {

// ---- image normalisation (dpvo.py:389) + f16 copy for the encoders: out = 2*(u8/255) - 0.5, same op order -------
__global__ void normalize_image_kernel(const uint8_t* __restrict__ img, float* __restrict__ f32, _Float16* __restrict__ f16,
                                       int64_t n) {
  // 16 pixels per thread and step (one 16-byte load, 32 / 64-byte stores) when the three pointers allow it: the kernel opens
  // the encoder chain of every frame, and as a byte-per-thread loop it was 3 600 workgroups of single-byte loads
  const bool vec = ((reinterpret_cast<uintptr_t>(img) | reinterpret_cast<uintptr_t>(f32) | reinterpret_cast<uintptr_t>(f16)) & 15) == 0;
  const int64_t nv = vec ? n / 16 : 0;
  for (int64_t q = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; q < nv; q += (int64_t)gridDim.x * blockDim.x) {
    const uint4 raw = reinterpret_cast<const uint4*>(img)[q];
    const uint32_t w[4] = {raw.x, raw.y, raw.z, raw.w};
    float v[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      // torch divides by a scalar as a multiplication by its f32 reciprocal (BinaryDivTrueKernel): match it bit for bit
      v[k] = 2.0f * ((float)((w[k >> 2] >> (8 * (k & 3))) & 0xffu) * (1.0f / 255.0f)) - 0.5f;
    }
    if (f32) {
#pragma unroll
      for (int k = 0; k < 4; ++k) reinterpret_cast<f4*>(f32)[4 * q + k] = (f4){v[4 * k], v[4 * k + 1], v[4 * k + 2], v[4 * k + 3]};
    }
    if (f16) {
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        h8 o;
#pragma unroll
        for (int i = 0; i < 8; ++i) o[i] = (_Float16)v[8 * k + i];
        reinterpret_cast<h8*>(f16)[2 * q + k] = o;
      }
    }
  }
  for (int64_t i = 16 * nv + blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const float v = 2.0f * ((float)img[i] * (1.0f / 255.0f)) - 0.5f;
    if (f32) f32[i] = v;
    if (f16) f16[i] = (_Float16)v;
  }
}

// ---- patch colours (dpvo.py:404-405 with net.py:143): colours_[n][m] = uint8((img_norm[c', y, x] + 0.5) * 127.5),
//      c' = (2,1,0), (x,y) = 4*(centroid + 0.5), zero when out of bounds; same float op order as the reference ------
__global__ void patch_colors_kernel(const uint8_t* __restrict__ img, const float* __restrict__ coords,
                                    uint8_t* __restrict__ out, int M, int H, int W) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= 3 * M) return;
  const int m = t / 3, c = t - 3 * m;
  const float x = 4.0f * (coords[2 * m + 0] + 0.5f), y = 4.0f * (coords[2 * m + 1] + 0.5f);
  const float fx = floorf(x), fy = floorf(y);
  const float dx = x - fx, dy = y - fy;
  const int j0 = (int)fx, i0 = (int)fy;
  const int cs = 2 - c;                                            // clr[0,:,[2,1,0]]
  float v[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      const int i = i0 + a, j = j0 + b;
      v[a][b] = (i >= 0 && i < H && j >= 0 && j < W) ? 2.0f * ((float)img[((int64_t)cs * H + i) * W + j] * (1.0f / 255.0f)) - 0.5f : 0.f;
    }
  const float o = blend4_ref(dx, dy, v[0][0], v[0][1], v[1][0], v[1][1]);
  const float f = (o + 0.5f) * (255.0f / 2.0f);
  out[t] = (uint8_t)(int)f;                                        // tensor.to(torch.uint8): truncation toward zero
}

// ---- feature ring-buffer store: fmap [C,h,w] (NCHW f16, the encoder output) -> channels-last slot [h,w,C] and the
//      4x4 average pool [h/4,w/4,C] (dpvo.py:437-438: F.avg_pool2d(fmap,1,1), F.avg_pool2d(fmap,4,4)) ------------
//      block = one 4x4-pooled row segment: 4 rows x 64 columns x all channels, transposed through LDS.
template <typename T>
__global__ __launch_bounds__(256) void store_features_kernel(const T* __restrict__ fmap, T* __restrict__ f1,
                                                             T* __restrict__ f2, int C, int h, int w) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  T* tile = reinterpret_cast<T*>(smem_raw);                        // [C][4][65]
  const int x0 = blockIdx.x * 64, y0 = blockIdx.y * 4;
  const int tid = threadIdx.x;
  const int nx = min(64, w - x0), ny = min(4, h - y0);
  for (int idx = tid; idx < C * 4 * 64; idx += 256) {
    const int x = idx & 63, y = (idx >> 6) & 3, c = idx >> 8;
    T v = (T)0.f;
    if (x < nx && y < ny) v = fmap[((int64_t)c * h + y0 + y) * w + x0 + x];
    tile[(c * 4 + y) * 65 + x] = v;
  }
  __syncthreads();
  for (int idx = tid; idx < 4 * 64 * C; idx += 256) {
    const int c = idx % C, x = (idx / C) & 63, y = idx / (C * 64);
    if (x < nx && y < ny) f1[(((int64_t)(y0 + y)) * w + x0 + x) * C + c] = tile[(c * 4 + y) * 65 + x];
  }
  if (ny == 4) {
    const int h4 = h / 4, w4 = w / 4;
    for (int idx = tid; idx < 16 * C; idx += 256) {
      const int c = idx % C, px = idx / C;
      if (x0 / 4 + px < w4 && px * 4 + 3 < nx && blockIdx.y < h4) {
        float s = 0.f;
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
          for (int b = 0; b < 4; ++b) s += (float)tile[(c * 4 + a) * 65 + px * 4 + b];
        f2[(((int64_t)blockIdx.y) * w4 + x0 / 4 + px) * C + c] = (T)(s / 16.0f);
      }
    }
  }
}

// ---- new edges of a frame (append_factors(edges_forw) + append_factors(edges_back), dpvo.py:215-221,362-375,458-459):
//      writes kk, jj, ii = ix[kk] for the n_forw + n_back new edges and zeroes their hidden state rows ---------------
__device__ __forceinline__ void append_edges_body(int64_t* __restrict__ ii, int64_t* __restrict__ jj, int64_t* __restrict__ kk,
                                                  float* __restrict__ net, const int64_t* __restrict__ ix, int64_t E0, int n,
                                                  int M, int r, int D, int64_t bid, int64_t nblk) {
  // forw: kk in [M*max(n-r,0), M*max(n-1,0)), jj = n-1 (kk-major);  back: kk in [M*(n-1), M*n) x jj in [max(n-r,0), n)
  const int64_t f0 = (int64_t)M * max(n - r, 0), f1 = (int64_t)M * max(n - 1, 0);
  const int64_t nf = f1 - f0;
  const int jlo = max(n - r, 0), nj = n - jlo;
  const int64_t nb = (int64_t)M * nj;
  const int64_t total = nf + nb;
  const int64_t gt = bid * (int64_t)blockDim.x + threadIdx.x;
  for (int64_t e = gt; e < total; e += nblk * blockDim.x) {
    int64_t k, j;
    if (e < nf) { k = f0 + e; j = n - 1; }
    else { const int64_t q = e - nf; k = (int64_t)M * max(n - 1, 0) + q / nj; j = jlo + q % nj; }
    kk[E0 + e] = k; jj[E0 + e] = j; ii[E0 + e] = ix[k];
  }
  if (!net) return;               // (hidden-state rows handled by the consumer: dpvo_update_forward_fused_rows reads new edges as zeros)
  const int64_t nn = total * (D / 4);
  f4* np = reinterpret_cast<f4*>(net + E0 * D);
  for (int64_t q = gt; q < nn; q += nblk * blockDim.x) np[q] = (f4)0.f;
}
__global__ void append_edges_kernel(int64_t* __restrict__ ii, int64_t* __restrict__ jj, int64_t* __restrict__ kk,
                                    float* __restrict__ net, const int64_t* __restrict__ ix, int64_t E0, int n, int M,
                                    int r, int D) {
  append_edges_body(ii, jj, kk, net, ix, E0, n, M, r, D, blockIdx.x, gridDim.x);
}

// ---- edge compaction: dst[t] = src[idx[t]] for the six per-edge arrays (remove_factors, dpvo.py:223-238) ----------
struct GatherArgs {
  const int64_t* idx; int64_t n;
  int64_t *oii, *ojj, *okk; float *onet, *otarget, *oweight;
};
__device__ __forceinline__ void gather_edges_body(const GatherArgs& G, const int64_t* __restrict__ ii, const int64_t* __restrict__ jj,
                                                  const int64_t* __restrict__ kk, const float* __restrict__ net,
                                                  const float* __restrict__ target, const float* __restrict__ weight, int D,
                                                  int64_t bid, int64_t nblk) {
  const int64_t gt = bid * (int64_t)blockDim.x + threadIdx.x, gs = nblk * blockDim.x;
  const int64_t* __restrict__ idx = G.idx;
  const int64_t n = G.n;
  for (int64_t t = gt; t < n; t += gs) {
    const int64_t s = idx[t];
    if (G.oii) { G.oii[t] = ii[s]; G.ojj[t] = jj[s]; G.okk[t] = kk[s]; }
    if (G.otarget) { G.otarget[2 * t] = target[2 * s]; G.otarget[2 * t + 1] = target[2 * s + 1]; }
    if (G.oweight) { G.oweight[2 * t] = weight[2 * s]; G.oweight[2 * t + 1] = weight[2 * s + 1]; }
  }
  if (G.onet) {
    const int dq = D / 4;
    for (int64_t q = gt; q < n * dq; q += gs) {
      const int64_t t = q / dq;
      const int c = (int)(q - t * dq);
      reinterpret_cast<f4*>(G.onet)[t * dq + c] = reinterpret_cast<const f4*>(net)[idx[t] * dq + c];
    }
  }
}
// one or two gather jobs from the same source arrays in one launch (blocks [0, nblkA) do job A, the rest job B)
__global__ void gather_edges_kernel(GatherArgs A, GatherArgs B, int nblkA, const int64_t* __restrict__ ii,
                                    const int64_t* __restrict__ jj, const int64_t* __restrict__ kk, const float* __restrict__ net,
                                    const float* __restrict__ target, const float* __restrict__ weight, int D) {
  if ((int)blockIdx.x < nblkA) gather_edges_body(A, ii, jj, kk, net, target, weight, D, blockIdx.x, nblkA);
  else gather_edges_body(B, ii, jj, kk, net, target, weight, D, blockIdx.x - nblkA, gridDim.x - nblkA);
}

// ---- motion model (dpvo.py:410-421): poses[n] = Exp(damping*fac * Log(P1 * P2^-1)) * P1 --------------------------
struct Q4 { float x, y, z, w; };
__device__ __forceinline__ Q4 qn(Q4 q) { const float n = sqrtf(q.x*q.x+q.y*q.y+q.z*q.z+q.w*q.w); return {q.x/n,q.y/n,q.z/n,q.w/n}; }
__device__ __forceinline__ Q4 qm(Q4 a, Q4 b) {
  return {a.w*b.x+a.x*b.w+a.y*b.z-a.z*b.y, a.w*b.y+a.y*b.w+a.z*b.x-a.x*b.z, a.w*b.z+a.z*b.w+a.x*b.y-a.y*b.x,
          a.w*b.w-a.x*b.x-a.y*b.y-a.z*b.z};
}
__device__ __forceinline__ void qr(Q4 q, const float* p, float* o) {
  float ux = q.y*p[2]-q.z*p[1], uy = q.z*p[0]-q.x*p[2], uz = q.x*p[1]-q.y*p[0];
  ux += ux; uy += uy; uz += uz;
  o[0] = p[0]+q.w*ux+(q.y*uz-q.z*uy); o[1] = p[1]+q.w*uy+(q.z*ux-q.x*uz); o[2] = p[2]+q.w*uz+(q.x*uy-q.y*ux);
}
__device__ __forceinline__ void hat3(const float* p, float* M) {
  M[0]=0; M[1]=-p[2]; M[2]=p[1]; M[3]=p[2]; M[4]=0; M[5]=-p[0]; M[6]=-p[1]; M[7]=p[0]; M[8]=0;
}
__device__ __forceinline__ void mm3(const float* A, const float* B, float* C) {
  for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) { float s = 0; for (int k = 0; k < 3; k++) s += A[3*i+k]*B[3*k+j]; C[3*i+j] = s; }
}
__device__ __forceinline__ void motion_model_body(float* __restrict__ poses, int n, float scale) {
  if (threadIdx.x != 0) return;
  constexpr float kEps = 1e-6f;
  const float* p1 = poses + 7 * (int64_t)(n - 1);
  const float* p2 = poses + 7 * (int64_t)(n - 2);
  // SE3 constructors normalise (so3.h:35-37); same sequence as lietorch: P2.inv(), P1 * inv, .log(), * scale, exp, * P1
  const Q4 q1 = qn({p1[3], p1[4], p1[5], p1[6]}), q2 = qn({p2[3], p2[4], p2[5], p2[6]});
  const Q4 q2i = qn({-q2.x, -q2.y, -q2.z, q2.w});
  float t2i[3]; qr(q2i, p2, t2i); t2i[0] = -t2i[0]; t2i[1] = -t2i[1]; t2i[2] = -t2i[2];
  const Q4 qd = qn(qm(q1, q2i));
  float td[3]; qr(q1, t2i, td); td[0] += p1[0]; td[1] += p1[1]; td[2] += p1[2];
  // log (se3.h:124-131, so3.h:115-150,192-208)
  const float sn = qd.x*qd.x+qd.y*qd.y+qd.z*qd.z, w = qd.w;
  float f;
  if (sn < kEps*kEps) f = 2.0f/w - (2.0f/3.0f)*sn/(w*w*w);
  else { const float nn = sqrtf(sn); f = (fabsf(w) < kEps) ? ((w > 0) ? 3.14159265358979323846f/nn : -3.14159265358979323846f/nn) : 2.0f*atanf(nn/w)/nn; }
  float phi[3] = {f*qd.x, f*qd.y, f*qd.z};
  float Phi[9], Phi2[9]; hat3(phi, Phi); mm3(Phi, Phi, Phi2);
  float th2 = phi[0]*phi[0]+phi[1]*phi[1]+phi[2]*phi[2], th = sqrtf(th2), ht = 0.5f*th;
  const float c2l = (th < kEps) ? (1.0f/12.0f) : (1.0f - th*cosf(ht)/(2.0f*sinf(ht)))/(th*th);
  float xi[6];
  for (int r = 0; r < 3; r++) { float s = 0; for (int c = 0; c < 3; c++) s += (((r==c)?1.0f:0.0f) - 0.5f*Phi[3*r+c] + c2l*Phi2[3*r+c])*td[c]; xi[r] = s; }
  xi[3] = phi[0]; xi[4] = phi[1]; xi[5] = phi[2];
  for (int r = 0; r < 6; r++) xi[r] = scale * xi[r];
  // exp (se3.h:133-142, so3.h:152-190)
  phi[0] = xi[3]; phi[1] = xi[4]; phi[2] = xi[5];
  th2 = phi[0]*phi[0]+phi[1]*phi[1]+phi[2]*phi[2]; th = sqrtf(th2);
  float imag, real;
  if (th < kEps) { const float t4 = th2*th2; imag = 0.5f-(1.0f/48.0f)*th2+(1.0f/3840.0f)*t4; real = 1.0f-(1.0f/8.0f)*th2+(1.0f/384.0f)*t4; }
  else { imag = sinf(.5f*th)/th; real = cosf(.5f*th); }
  const Q4 qe = qn({imag*phi[0], imag*phi[1], imag*phi[2], real});
  hat3(phi, Phi); mm3(Phi, Phi, Phi2);
  const float c1 = (th < kEps) ? 0.5f-(1.0f/24.0f)*th2 : (1.0f-cosf(th))/th2;
  const float c2 = (th < kEps) ? (1.0f/6.0f)-(1.0f/120.0f)*th2 : (th-sinf(th))/(th2*th);
  float te[3];
  for (int r = 0; r < 3; r++) { float s = 0; for (int c = 0; c < 3; c++) s += (((r==c)?1.0f:0.0f) + c1*Phi[3*r+c] + c2*Phi2[3*r+c])*xi[c]; te[r] = s; }
  // exp(xi) * P1
  const Q4 qo = qn(qm(qe, q1));
  float to[3]; qr(qe, p1, to);
  float* o = poses + 7 * (int64_t)n;
  o[0] = te[0]+to[0]; o[1] = te[1]+to[1]; o[2] = te[2]+to[2]; o[3] = qo.x; o[4] = qo.y; o[5] = qo.z; o[6] = qo.w;
}
__global__ void motion_model_kernel(float* __restrict__ poses, int n, float scale) {
  if (blockIdx.x == 0) motion_model_body(poses, n, scale);
}

// ---- depth initialisation (dpvo.py:427-432): patches[n][:, 2] = median(patches[n-3:n, :, 2]) (torch.median = lower
//      median of the flattened values).  Rank counting instead of a sort: element i is the lower median iff exactly
//      (cnt-1)/2 elements precede it in the total order (value, index).  kMedPerBlock = 4 elements per block, ONE WAVE per
//      element: lane l compares against candidates 4 (l + 64 k) .. + 3 (conflict-free 16-byte LDS reads, ~10 trips for the
//      usual 2 592 values; with 8 lanes per element the 81 trips of a lane were 17 us of the frame's start); the one block
//      that owns the median writes the new frame.
constexpr int kMedPerBlock = 4;
__device__ __forceinline__ void median_depth_body(float* __restrict__ patches, int n, int M, int PP, int bid) {
  __shared__ __attribute__((aligned(16))) float v[4096 + 32];
  __shared__ float med_s;
  __shared__ int found_s;
  const int per = M * PP, cnt = 3 * per;
  const float* src = patches + (int64_t)(n - 3) * M * 3 * PP;
  if (threadIdx.x == 0) found_s = 0;
  for (int i = threadIdx.x; i < cnt; i += 256) {
    const int f = i / per, r = i - f * per, m = r / PP, p = r - m * PP;
    v[i] = src[((int64_t)(f * M + m) * 3 + 2) * PP + p];
  }
  __syncthreads();
  const int lane = threadIdx.x & 63;
  const int i = bid * kMedPerBlock + (threadIdx.x >> 6);
  const float x = i < cnt ? v[i] : 0.f;
  int rank = 0;
  for (int j = 4 * lane; j < cnt; j += 256) {
    const float4 y = *reinterpret_cast<const float4*>(v + j);
    rank += (y.x < x) || (y.x == x && j < i);
    if (j + 1 < cnt) rank += (y.y < x) || (y.y == x && j + 1 < i);
    if (j + 2 < cnt) rank += (y.z < x) || (y.z == x && j + 2 < i);
    if (j + 3 < cnt) rank += (y.w < x) || (y.w == x && j + 3 < i);
  }
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) rank += __shfl_xor(rank, o);
  if (lane == 0 && i < cnt && rank == (cnt - 1) / 2) { med_s = x; found_s = 1; }
  __syncthreads();
  if (!found_s) return;
  const float med = med_s;
  float* dst = patches + (int64_t)n * M * 3 * PP;
  for (int k = threadIdx.x; k < per; k += 256) { const int m = k / PP, p = k - m * PP; dst[((int64_t)m * 3 + 2) * PP + p] = med; }
}
__global__ __launch_bounds__(256) void median_depth_kernel(float* __restrict__ patches, int n, int M, int PP) {
  median_depth_body(patches, n, M, PP, blockIdx.x);
}

// ---- everything a new frame contributes besides its feature maps, in one launch (Patchifier.forward's gathers
//      net.py:136-147 + the state stores of dpvo.py:401-438): per patch m
//        gmap_slot[m][a][b][c]  = patchify(fmap, coords, 1)        (bilinear blend of the floor-gathered 4x4 window)
//        imap_slot[m][c]        = patchify(imap, coords, 0)
//        patches_slot[m]        = patchify((x, y, 1) grid, coords, 1), plane 2 overwritten with depth[m]
//        colors_slot[m]         = patchify(image, 4 (coords + 0.5), 0) in RGB order, uint8
//      plus intrinsics_slot = intrinsics / res, index_row[:] = frame + 1, index_map = m_next.
//      coords come either as float [M,2] or as the two int64 randint draws (x, y) of net.py:132-133. -----------------
__device__ __forceinline__ int fp_floor_int(float v) {
  float f = floorf(v);
  if (!(f > -1.0e6f)) f = -1.0e6f;   // also catches NaN
  if (f > 1.0e6f) f = 1.0e6f;
  return (int)f;
}
template <typename F>
__device__ __forceinline__ float fp_blend(float x, float y, int i0, int j0, int H, int W, F at) {
  const float dx = x - floorf(x), dy = y - floorf(y);
  float v[2][2];
#pragma unroll
  for (int aa = 0; aa < 2; ++aa)
#pragma unroll
    for (int bb = 0; bb < 2; ++bb) {
      const int i = i0 + aa, j = j0 + bb;
      v[aa][bb] = (i >= 0 && i < H && j >= 0 && j < W) ? at(i, j) : 0.f;
    }
  const float o = blend4_ref(dx, dy, v[0][0], v[0][1], v[1][0], v[1][1]);
  return o;
}
struct FramePatchesArgs {
  const _Float16 *fmap, *imap; const uint8_t* img; const float* coords; const int64_t *xs, *ys; const float *depth, *intr;
  float res; _Float16 *gmap_slot, *imap_slot; float* patches_slot; uint8_t* colors_slot; float* intr_slot;
  int64_t *index_row, *index_map; float* coords_out; int M, h, w, H, W, CF, CI; int64_t frame_next, m_next;
  int skip_depth;          // the depth plane is written by someone else (the median of the previous frames)
};
__device__ __forceinline__ void frame_patches_body(const FramePatchesArgs& A, int m) {
  const _Float16 *fmap = A.fmap, *imap = A.imap; const uint8_t* img = A.img; const float* coords = A.coords;
  const int64_t *xs = A.xs, *ys = A.ys; const float *depth = A.depth, *intr = A.intr; const float res = A.res;
  _Float16 *gmap_slot = A.gmap_slot, *imap_slot = A.imap_slot; float* patches_slot = A.patches_slot;
  uint8_t* colors_slot = A.colors_slot; float* intr_slot = A.intr_slot; int64_t *index_row = A.index_row, *index_map = A.index_map;
  float* coords_out = A.coords_out; const int h = A.h, w = A.w, H = A.H, W = A.W, CF = A.CF, CI = A.CI;
  const int64_t frame_next = A.frame_next, m_next = A.m_next;
  const int t = threadIdx.x;
  const float x = coords ? coords[2 * m] : (float)xs[m], y = coords ? coords[2 * m + 1] : (float)ys[m];
  const int fi = fp_floor_int(y), fj = fp_floor_int(x);
  // gmap: 9 window positions x CF channels (channels fastest in both the NHWC source and the channels-last slot)
  // (every output group is optional: a null slot skips it, so the state stores and the feature gathers can be two launches)
  if (gmap_slot && imap_slot && !(CF % 8) && !(CI % 8) && 9 * (CF / 8) + CI / 8 <= 256) {
    // 8 channels (16 bytes) per thread: work items [0, 9 CF/8) = gmap (window position ab, channel group), then CI/8 imap groups;
    // the four taps of an item are four 16-byte loads in flight together.  (One channel per thread: 18 two-byte loads per
    // thread in five dependent trips, 7-10 us of part 2.)  Same blend, same rounding per channel.
    const int ng = 9 * (CF / 8), item = t;
    if (item < ng + CI / 8) {
      const bool isg = item < ng;
      const int grp = isg ? item % (CF / 8) : item - ng, ab = isg ? item / (CF / 8) : 4, a = ab / 3, b = ab - 3 * a;
      const int C = isg ? CF : CI, i0 = fi + a - 1, j0 = fj + b - 1;           // (imap: the centre position, ab = 4)
      const _Float16* src = isg ? fmap : imap;
      const float dx = x - floorf(x), dy = y - floorf(y);
      h8 tap[2][2];
#pragma unroll
      for (int aa = 0; aa < 2; ++aa)
#pragma unroll
        for (int bb = 0; bb < 2; ++bb) {
          const int i = i0 + aa, j = j0 + bb;
          if (i >= 0 && i < h && j >= 0 && j < w) tap[aa][bb] = *reinterpret_cast<const h8*>(src + ((int64_t)i * w + j) * C + 8 * grp);
          else tap[aa][bb] = (h8)(_Float16)0.f;
        }
      h8 o;
#pragma unroll
      for (int c = 0; c < 8; ++c)
        o[c] = (_Float16)blend4_ref(dx, dy, (float)tap[0][0][c], (float)tap[0][1][c], (float)tap[1][0][c], (float)tap[1][1][c]);
      _Float16* dst = isg ? gmap_slot + ((int64_t)m * 9 + ab) * CF + 8 * grp : imap_slot + (int64_t)m * CI + 8 * grp;
      *reinterpret_cast<h8*>(dst) = o;
    }
  } else {
  for (int e = t; gmap_slot && e < 9 * CF; e += 256) {
    const int c = e % CF, ab = e / CF, a = ab / 3, b = ab - 3 * a;
    const float o = fp_blend(x, y, fi + a - 1, fj + b - 1, h, w,
                             [&](int i, int j) { return (float)fmap[((int64_t)i * w + j) * CF + c]; });
    gmap_slot[((int64_t)m * 9 + ab) * CF + c] = (_Float16)o;
  }
  for (int c = t; imap_slot && c < CI; c += 256) {
    const float o = fp_blend(x, y, fi, fj, h, w, [&](int i, int j) { return (float)imap[((int64_t)i * w + j) * CI + c]; });
    imap_slot[(int64_t)m * CI + c] = (_Float16)o;
  }
  }
  if (t < 27) {
    if (!patches_slot) return;
    const int pl = t / 9, ab = t - 9 * pl, a = ab / 3, b = ab - 3 * a;
    float o;
    if (pl == 2) { if (A.skip_depth) return; o = depth[m]; }
    else o = fp_blend(x, y, fi + a - 1, fj + b - 1, h, w, [&](int i, int j) { return pl == 0 ? (float)j : (float)i; });
    patches_slot[(int64_t)m * 27 + t] = o;
  } else if (t >= 32 && t < 35) {
    if (!colors_slot) return;
    const int c = t - 32, cs = 2 - c;                                            // clr[0,:,[2,1,0]]
    const float cx = 4.0f * (x + 0.5f), cy = 4.0f * (y + 0.5f);
    const float o = fp_blend(cx, cy, (int)floorf(cy), (int)floorf(cx), H, W, [&](int i, int j) {
      return 2.0f * ((float)img[((int64_t)cs * H + i) * W + j] * (1.0f / 255.0f)) - 0.5f;
    });
    const float f = (o + 0.5f) * (255.0f / 2.0f);
    colors_slot[m * 3 + c] = (uint8_t)(int)f;
  } else if (t >= 64 && t < 66 && coords_out) {
    coords_out[2 * m + (t - 64)] = t == 64 ? x : y;
  }
  if (m == 0) {
    if (t >= 96 && t < 100 && intr_slot) intr_slot[t - 96] = intr[t - 96] / res;
    if (t == 100 && index_map) *index_map = m_next;
  }
  if (index_row && t == 128) index_row[m] = frame_next;
}
__global__ __launch_bounds__(256) void frame_patches_kernel(FramePatchesArgs A) { frame_patches_body(A, blockIdx.x); }

// dpvo_frame_state as ONE launch: the five jobs are independent of each other (the depth plane of the new frame belongs to
// the median job when that one runs), so their workgroups share a grid and run side by side instead of as a chain of
// five dependent launches.  Roles by block index: [0, M) patches | 1 motion model | n_med median | n_pool pyramid | rest edges.
struct FrameStateArgs {
  FramePatchesArgs fp;
  float* poses; int mm_n; float mm_scale;
  float* patches_all; int md_n, P;
  _Float16* fmap2_slot;
  int64_t *ii, *jj, *kk; float* net; const int64_t* ix; int64_t E0; int ap_n, ap_r, D;
  int n_med, n_pool, n_app;
  int32_t* clear_ptr; int clear_n;      // an int region some other launch wants cleared (the graph plan's counters): blocks behind the rest
};
#ifdef FS_TRACE
// instrumentation build (tools/fs_trace.sh): start / end of every workgroup of the last launch (100 MHz wall clock)
__device__ unsigned long long fs_trace_buf[2][4096][2];
#endif
__device__ __forceinline__ void frame_state_roles(const FrameStateArgs& S) {
  int b = blockIdx.x;
  if (b < S.fp.M) { frame_patches_body(S.fp, b); return; }
  b -= S.fp.M;
  if (b < 1) { if (S.poses) motion_model_body(S.poses, S.mm_n, S.mm_scale); return; }
  b -= 1;
  if (b < S.n_med) { median_depth_body(S.patches_all, S.md_n, S.fp.M, S.P * S.P, b); return; }
  b -= S.n_med;
  if (b < S.n_pool) { pool4_nhwc_body(S.fp.fmap, S.fmap2_slot, S.fp.h, S.fp.w, S.fp.CF, b, S.n_pool); return; }
  b -= S.n_pool;
  if (b < S.n_app) { append_edges_body(S.ii, S.jj, S.kk, S.net, S.ix, S.E0, S.ap_n, S.fp.M, S.ap_r, S.D, b, S.n_app); return; }
  b -= S.n_app;
  const int i = b * 256 + threadIdx.x;
  if (i < S.clear_n) S.clear_ptr[i] = 0;
}
__global__ __launch_bounds__(256) void frame_state_kernel(FrameStateArgs S) {
#ifdef FS_TRACE
  const int part = S.n_pool ? 1 : 0;
  if (threadIdx.x == 0 && blockIdx.x < 4096) fs_trace_buf[part][blockIdx.x][0] = wall_clock64();
#endif
  frame_state_roles(S);
#ifdef FS_TRACE
  __syncthreads();
  if (threadIdx.x == 0 && blockIdx.x < 4096) fs_trace_buf[part][blockIdx.x][1] = wall_clock64();
#endif
}
#ifdef FS_TRACE
}  // namespace
extern "C" int dpvo_debug_fs_trace(unsigned long long* out) {
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(fs_trace_buf), sizeof(fs_trace_buf)) == hipSuccess ? 0 : 1;
}
namespace {
#endif

inline unsigned grid_for(int64_t n, int cap = 4096) {
  int64_t g = cdiv64(n, 256);
  return (unsigned)(g < 1 ? 1 : (g > cap ? cap : g));
}

}  // namespace

extern "C" int dpvo_normalize_image(const void* img_u8, float* out_f32, void* out_f16, int64_t n, void* stream) {
  if (n < 0 || (!out_f32 && !out_f16)) return DPVO_E_INVALID;
  if (n == 0) return DPVO_OK;
  if (!img_u8) return DPVO_E_INVALID;
  hipLaunchKernelGGL(normalize_image_kernel, dim3(grid_for((n + 15) / 16)), dim3(256), 0, (hipStream_t)stream, (const uint8_t*)img_u8,
                     out_f32, (_Float16*)out_f16, n);
  DPVO_LAUNCH_CHECK();
  return DPVO_OK;
}

extern "C" int dpvo_patch_colors(const void* img_u8, const float* coords, void* out_u8, int M, int H, int W, void* stream) {
  if (M < 0 || H <= 0 || W <= 0) return DPVO_E_INVALID;
  if (M == 0) return DPVO_OK;
  if (!img_u8 || !coords || !out_u8) return DPVO_E_INVALID;
  hipLaunchKernelGGL(patch_colors_kernel, dim3((3 * M + 255) / 256), dim3(256), 0, (hipStream_t)stream,
                     (const uint8_t*)img_u8, coords, (uint8_t*)out_u8, M, H, W);
  DPVO_LAUNCH_CHECK();
  return DPVO_OK;
}

extern "C" int dpvo_store_features(const void* fmap, void* f1_slot, void* f2_slot, int dtype, int C, int h, int w,
                                   void* stream) {
  if (C <= 0 || h <= 0 || w <= 0 || (h % 4) || (w % 4)) return DPVO_E_INVALID;
  if (!fmap || !f1_slot || !f2_slot) return DPVO_E_INVALID;
  const dim3 grid((w + 63) / 64, h / 4);
  if (dtype == DPVO_F16) {
    const size_t sh = (size_t)C * 4 * 65 * 2;
    if (sh > 160 * 1024) return DPVO_E_UNSUPPORTED;
    (void)hipFuncSetAttribute((const void*)store_features_kernel<_Float16>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh);
    hipLaunchKernelGGL(store_features_kernel<_Float16>, grid, dim3(256), sh, (hipStream_t)stream, (const _Float16*)fmap,
                       (_Float16*)f1_slot, (_Float16*)f2_slot, C, h, w);
  } else if (dtype == DPVO_F32) {
    const size_t sh = (size_t)C * 4 * 65 * 4;
    if (sh > 160 * 1024) return DPVO_E_UNSUPPORTED;
    (void)hipFuncSetAttribute((const void*)store_features_kernel<float>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh);
    hipLaunchKernelGGL(store_features_kernel<float>, grid, dim3(256), sh, (hipStream_t)stream, (const float*)fmap,
                       (float*)f1_slot, (float*)f2_slot, C, h, w);
  } else {
    return DPVO_E_UNSUPPORTED;
  }
  DPVO_LAUNCH_CHECK();
  return DPVO_OK;
}

extern "C" int dpvo_append_edges(int64_t* ii, int64_t* jj, int64_t* kk, float* net, const int64_t* ix, int64_t E0, int n,
                                 int M, int r, int D, int64_t* n_new, void* stream) {
  if (E0 < 0 || n < 1 || M <= 0 || r < 0 || D <= 0 || (D % 4)) return DPVO_E_INVALID;
  const int64_t nf = (int64_t)M * ((n - 1 > 0 ? n - 1 : 0) - (n - r > 0 ? n - r : 0));
  const int jlo = n - r > 0 ? n - r : 0;
  const int64_t total = nf + (int64_t)M * (n - jlo);
  if (n_new) *n_new = total;
  if (total == 0) return DPVO_OK;
  if (!ii || !jj || !kk || !ix) return DPVO_E_INVALID;       // (net == NULL: the caller treats the new state rows as zeros itself)
  hipLaunchKernelGGL(append_edges_kernel, dim3(grid_for(total * (D / 4), 2048)), dim3(256), 0, (hipStream_t)stream, ii, jj,
                     kk, net, ix, E0, n, M, r, D);
  DPVO_LAUNCH_CHECK();
  return DPVO_OK;
}

extern "C" int dpvo_gather_edges(const int64_t* idx, int64_t n, const int64_t* ii, const int64_t* jj, const int64_t* kk,
                                 const float* net, const float* target, const float* weight, int64_t* oii, int64_t* ojj,
                                 int64_t* okk, float* onet, float* otarget, float* oweight, int D, void* stream) {
  if (n < 0 || D <= 0 || (D % 4)) return DPVO_E_INVALID;
  if (n == 0) return DPVO_OK;
  if (!idx || !ii || !jj || !kk) return DPVO_E_INVALID;
  // (oii = ojj = okk = NULL with onet set: the hidden-state rows only -- a deferred compaction applied late)
  if ((!oii || !ojj || !okk) && (oii || ojj || okk || !onet || !net)) return DPVO_E_INVALID;
  const GatherArgs A = {idx, n, oii, ojj, okk, onet, otarget, oweight};
  const unsigned g = grid_for(onet ? n * (D / 4) : n, 2048);
  hipLaunchKernelGGL(gather_edges_kernel, dim3(g), dim3(256), 0, (hipStream_t)stream, A, A, (int)g, ii, jj, kk, net, target, weight, D);
  DPVO_LAUNCH_CHECK();
  return DPVO_OK;
}

extern "C" int dpvo_gather_edges2(const int64_t* idx_a, int64_t n_a, int64_t* a_ii, int64_t* a_jj, int64_t* a_kk, float* a_net,
                                  float* a_target, float* a_weight, const int64_t* idx_b, int64_t n_b, int64_t* b_ii,
                                  int64_t* b_jj, int64_t* b_kk, float* b_net, float* b_target, float* b_weight,
                                  const int64_t* ii, const int64_t* jj, const int64_t* kk, const float* net,
                                  const float* target, const float* weight, int D, void* stream) {
  if (n_a < 0 || n_b < 0 || D <= 0 || (D % 4)) return DPVO_E_INVALID;
  if (n_a == 0) return dpvo_gather_edges(idx_b, n_b, ii, jj, kk, net, target, weight, b_ii, b_jj, b_kk, b_net, b_target, b_weight, D, stream);
  if (n_b == 0) return dpvo_gather_edges(idx_a, n_a, ii, jj, kk, net, target, weight, a_ii, a_jj, a_kk, a_net, a_target, a_weight, D, stream);
  if (!idx_a || !idx_b || !ii || !jj || !kk || !a_ii || !a_jj || !a_kk || !b_ii || !b_jj || !b_kk) return DPVO_E_INVALID;
  const GatherArgs A = {idx_a, n_a, a_ii, a_jj, a_kk, a_net, a_target, a_weight}, B = {idx_b, n_b, b_ii, b_jj, b_kk, b_net, b_target, b_weight};
  const unsigned ga = grid_for(a_net ? n_a * (D / 4) : n_a, 2048), gb = grid_for(b_net ? n_b * (D / 4) : n_b, 2048);
  hipLaunchKernelGGL(gather_edges_kernel, dim3(ga + gb), dim3(256), 0, (hipStream_t)stream, A, B, (int)ga, ii, jj, kk, net, target,
                     weight, D);
  DPVO_LAUNCH_CHECK();
  return DPVO_OK;
}

extern "C" int dpvo_frame_patches(const void* fmap, const void* imap, const void* img_u8, const float* coords,
                                  const int64_t* xs, const int64_t* ys, const float* depth, const float* intrinsics,
                                  float res, void* gmap_slot, void* imap_slot, float* patches_slot, void* colors_slot,
                                  float* intrinsics_slot, int64_t* index_row, int64_t* index_map, float* coords_out, int M,
                                  int h, int w, int H, int W, int CF, int CI, int P, int64_t frame_next, int64_t m_next,
                                  void* stream) {
  if (M < 0 || h <= 0 || w <= 0 || H <= 0 || W <= 0 || CF <= 0 || CI <= 0) return DPVO_E_INVALID;
  if (P != 3) return DPVO_E_UNSUPPORTED;
  if (M == 0) return DPVO_OK;
  // output groups are optional (null = skip); each needs its inputs
  if (!gmap_slot && !imap_slot && !patches_slot && !colors_slot) return DPVO_E_INVALID;
  if ((gmap_slot && !fmap) || (imap_slot && !imap) || (patches_slot && !depth) || (colors_slot && !img_u8)) return DPVO_E_INVALID;
  if (!coords && !(xs && ys)) return DPVO_E_INVALID;
  if (intrinsics_slot && !intrinsics) return DPVO_E_INVALID;
  const FramePatchesArgs A = {(const _Float16*)fmap, (const _Float16*)imap, (const uint8_t*)img_u8, coords, xs, ys, depth, intrinsics,
                              res, (_Float16*)gmap_slot, (_Float16*)imap_slot, patches_slot, (uint8_t*)colors_slot, intrinsics_slot,
                              index_row, index_map, coords_out, M, h, w, H, W, CF, CI, frame_next, m_next, 0};
  hipLaunchKernelGGL(frame_patches_kernel, dim3(M), dim3(256), 0, (hipStream_t)stream, A);
  DPVO_LAUNCH_CHECK();
  return DPVO_OK;
}

// part 0: everything (dpvo_frame_state).  part 1: what does NOT read the encoders' outputs -- coordinate / depth patches, colours,
// intrinsics, index rows, motion model, depth median, the new edges; part 2: what does -- the gmap / imap gathers and pyramid
// level 1.  dpvo_frame_update issues part 1, the plan and the reprojection BEFORE it makes its stream wait for the side stream's
// encoders, and part 2 behind that wait: ~45 us of small launches that no longer sit between "encoders done" and the correlation.
extern "C" int dpvo_frame_state_part(dpvo_frame_state_t* p, int part, void* stream) {
  return dpvo_frame_state_part_clear(p, part, nullptr, 0, stream);
}
// (internal, csrc/common.h)  the same; part 1 also clears clear_count ints at clear_ptr with a few extra workgroups
extern "C" int dpvo_frame_state_part_clear(dpvo_frame_state_t* p, int part, int32_t* clear_ptr, int64_t clear_count, void* stream) {
  if (!p || part < 0 || part > 2) return DPVO_E_INVALID;
  if (clear_ptr && (part != 1 || clear_count <= 0 || clear_count > (1 << 24))) return DPVO_E_INVALID;
  if (part == 0) return dpvo_frame_state(p, stream);
  dpvo_frame_state_t q = *p;
  if (part == 1) { q.gmap_slot = q.imap_slot = q.fmap2_slot = nullptr; }
  else { q.patches_slot = nullptr; q.colors_slot = nullptr; q.intrinsics_slot = nullptr; q.index_row = nullptr; q.index_map = nullptr;
         q.poses = nullptr; q.patches_all = nullptr; q.ii = nullptr; }
  const int M = q.M;
  if (M <= 0 || q.P != 3 || q.h <= 0 || q.w <= 0 || (q.h % 4) || (q.w % 4) || (q.CF % 8) || q.CI <= 0 || !(q.coords || (q.xs && q.ys)))
    return DPVO_E_INVALID;
  if (part == 2 && (!q.fmap || !q.imap || !q.gmap_slot || !q.imap_slot || !q.fmap2_slot)) return DPVO_E_INVALID;
  if (part == 1 && (!q.img_u8 || !q.patches_slot || !q.colors_slot || !q.poses || !q.patches_all || !q.ii || !q.jj || !q.kk || !q.ix ||
                    q.mm_n < 2 || q.md_n < 3 || 3 * M * 9 > 4096 || q.E0 < 0 || q.ap_n < 1 || q.ap_r < 0 || q.D <= 0 || (q.D % 4)))
    return DPVO_E_INVALID;
  FrameStateArgs S;
  S.fp = {(const _Float16*)q.fmap, (const _Float16*)q.imap, (const uint8_t*)q.img_u8, q.coords, q.xs, q.ys, q.depth,
          q.intrinsics, q.res, (_Float16*)q.gmap_slot, (_Float16*)q.imap_slot, q.patches_slot, (uint8_t*)q.colors_slot,
          q.intrinsics_slot, q.index_row, q.index_map, nullptr, M, q.h, q.w, q.H, q.W, q.CF, q.CI, q.frame_next, q.m_next, 1};
  S.poses = q.poses; S.mm_n = q.mm_n; S.mm_scale = q.mm_scale;
  S.patches_all = q.patches_all; S.md_n = q.md_n; S.P = q.P;
  S.fmap2_slot = (_Float16*)q.fmap2_slot;
  S.ii = q.ii; S.jj = q.jj; S.kk = q.kk; S.net = q.net; S.ix = q.ix; S.E0 = q.E0; S.ap_n = q.ap_n; S.ap_r = q.ap_r; S.D = q.D;
  S.n_med = 0; S.n_pool = 0; S.n_app = 0;
  S.clear_ptr = clear_ptr; S.clear_n = clear_ptr ? (int)clear_count : 0;
  const int n_clr = (S.clear_n + 255) / 256;
  if (part == 1) {
    const int n = q.ap_n, r = q.ap_r, jlo = n - r > 0 ? n - r : 0;
    const int64_t total = (int64_t)M * ((n - 1 > 0 ? n - 1 : 0) - jlo) + (int64_t)M * (n - jlo);
    p->n_new = total;
    S.n_med = (3 * M * 9 + kMedPerBlock - 1) / kMedPerBlock;
    S.n_app = total > 0 ? (int)grid_for(total * (q.D / 4), 2048) : 0;
  } else {
    S.n_pool = (int)(((int64_t)(q.h / 4) * (q.w / 4) * (q.CF / 8) + 255) / 256);
  }
  hipLaunchKernelGGL(frame_state_kernel, dim3((unsigned)(M + 1 + S.n_med + S.n_pool + S.n_app + n_clr)), dim3(256), 0, (hipStream_t)stream, S);
  DPVO_LAUNCH_CHECK();
  return DPVO_OK;
}

extern "C" int dpvo_frame_state(dpvo_frame_state_t* p, void* stream) {
  if (!p) return DPVO_E_INVALID;
  const bool fused = p->poses && p->patches_all && p->fmap2_slot && p->ii && p->jj && p->kk && p->ix && p->fmap && p->imap &&
                     p->img_u8 && p->gmap_slot && p->imap_slot && p->patches_slot && p->colors_slot && (p->coords || (p->xs && p->ys)) &&
                     (!p->intrinsics_slot || p->intrinsics) && p->P == 3 && p->M > 0 && p->mm_n >= 2 && p->md_n >= 3 &&
                     3 * p->M * 9 <= 4096 && p->h > 0 && p->w > 0 && !(p->h % 4) && !(p->w % 4) && !(p->CF % 8) && p->CI > 0 &&
                     p->H > 0 && p->W > 0 && p->E0 >= 0 && p->ap_n >= 1 && p->ap_r >= 0 && p->D > 0 && !(p->D % 4);
  if (fused) {
    const int n = p->ap_n, r = p->ap_r, M = p->M;
    const int jlo = n - r > 0 ? n - r : 0;
    const int64_t total = (int64_t)M * ((n - 1 > 0 ? n - 1 : 0) - jlo) + (int64_t)M * (n - jlo);
    p->n_new = total;
    FrameStateArgs S;
    S.fp = {(const _Float16*)p->fmap, (const _Float16*)p->imap, (const uint8_t*)p->img_u8, p->coords, p->xs, p->ys, p->depth,
            p->intrinsics, p->res, (_Float16*)p->gmap_slot, (_Float16*)p->imap_slot, p->patches_slot, (uint8_t*)p->colors_slot,
            p->intrinsics_slot, p->index_row, p->index_map, nullptr, M, p->h, p->w, p->H, p->W, p->CF, p->CI, p->frame_next,
            p->m_next, 1};
    S.poses = p->poses; S.mm_n = p->mm_n; S.mm_scale = p->mm_scale;
    S.patches_all = p->patches_all; S.md_n = p->md_n; S.P = p->P;
    S.fmap2_slot = (_Float16*)p->fmap2_slot;
    S.ii = p->ii; S.jj = p->jj; S.kk = p->kk; S.net = p->net; S.ix = p->ix; S.E0 = p->E0; S.ap_n = n; S.ap_r = r; S.D = p->D;
    S.n_med = (3 * M * 9 + kMedPerBlock - 1) / kMedPerBlock;
    S.n_pool = (int)(((int64_t)(p->h / 4) * (p->w / 4) * (p->CF / 8) + 255) / 256);
    S.n_app = total > 0 ? (int)grid_for(total * (p->D / 4), 2048) : 0;
    S.clear_ptr = nullptr; S.clear_n = 0;
    hipLaunchKernelGGL(frame_state_kernel, dim3((unsigned)(M + 1 + S.n_med + S.n_pool + S.n_app)), dim3(256), 0, (hipStream_t)stream,
                       S);
    DPVO_LAUNCH_CHECK();
    return DPVO_OK;
  }
  int rc = dpvo_frame_patches(p->fmap, p->imap, p->img_u8, p->coords, p->xs, p->ys, p->depth, p->intrinsics, p->res,
                              p->gmap_slot, p->imap_slot, p->patches_slot, p->colors_slot, p->intrinsics_slot, p->index_row,
                              p->index_map, nullptr, p->M, p->h, p->w, p->H, p->W, p->CF, p->CI, p->P, p->frame_next, p->m_next,
                              stream);
  if (rc == DPVO_OK && p->poses) rc = dpvo_motion_model(p->poses, p->mm_n, p->mm_scale, stream);
  if (rc == DPVO_OK && p->patches_all) rc = dpvo_median_depth(p->patches_all, p->md_n, p->M, p->P, stream);
  if (rc == DPVO_OK && p->fmap2_slot) rc = dpvo_pool4_nhwc(p->fmap, p->fmap2_slot, p->h, p->w, p->CF, stream);
  if (rc == DPVO_OK && p->ii) rc = dpvo_append_edges(p->ii, p->jj, p->kk, p->net, p->ix, p->E0, p->ap_n, p->M, p->ap_r, p->D,
                                                    &p->n_new, stream);
  return rc;
}

extern "C" int dpvo_motion_model(float* poses, int n, float scale, void* stream) {
  if (!poses || n < 2) return DPVO_E_INVALID;
  hipLaunchKernelGGL(motion_model_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, poses, n, scale);
  DPVO_LAUNCH_CHECK();
  return DPVO_OK;
}

extern "C" int dpvo_median_depth(float* patches, int n, int M, int P, void* stream) {
  if (!patches || n < 3 || M <= 0 || P <= 0) return DPVO_E_INVALID;
  if (3 * M * P * P > 4096) return DPVO_E_UNSUPPORTED;
  hipLaunchKernelGGL(median_depth_kernel, dim3((3 * M * P * P + kMedPerBlock - 1) / kMedPerBlock), dim3(256), 0, (hipStream_t)stream, patches, n, M,
                     P * P);
  DPVO_LAUNCH_CHECK();
  return DPVO_OK;
}

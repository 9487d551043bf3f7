// pipe_probe.hip -- does a software pipeline across two row tiles pay in update_fused.hip's kernels?
//   seq  : what the product kernels do: GEMM of a 32 RT-row tile (24 k-steps, weights streamed from L2 into a DW-deep ring, B fragments
//          from LDS), THEN its epilogue (f32 -> f16, activation, ds_write_b128 into the tile), barrier, next layer.
//   pipe : two tiles A, B of 32 RT rows each per workgroup; the k-loop of one tile's GEMM carries, in the shadow of its MFMAs, a
//          1/24 slice of the OTHER tile's epilogue per k-step (4 of its 48 RT values per lane): GEMM_i(A) || epi_{i-1}(B), barrier,
//          GEMM_i(B) || epi_i(A), barrier.  Same arithmetic per row; the weights of a layer are streamed twice (once per tile).
// ACT: 1 relu (to_lds<1>), 2 sigmoid (to_lds<2>: v_exp + v_rcp per value, K7's heaviest epilogue).
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/pipe_probe tools/probes/pipe_probe.hip && /tmp/pipe_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));
constexpr int PITCH = 784;

template <int ACT>
__device__ __forceinline__ _Float16 actf(float v) {
  _Float16 x = (_Float16)v;
  if (ACT == 1) x = x > (_Float16)0 ? x : (_Float16)0;
  if (ACT == 2) x = (_Float16)__builtin_amdgcn_rcpf(1.0f + __expf(-(float)x));
  return x;
}

// the whole epilogue of one tile (as in update_fused_dev.h: to_lds)
template <int RT, int ACT>
__device__ __forceinline__ void epi_all(const f16v (&v)[RT][3], char* al, int w) {
#pragma unroll
  for (int r = 0; r < RT; ++r)
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        h8 o;
#pragma unroll
        for (int i = 0; i < 8; ++i) o[i] = actf<ACT>(v[r][t][8 * c + i]);
        *reinterpret_cast<h8*>(al + r * 32 * PITCH + ((3 * w + t) * 2 + c) * 32) = o;
      }
}
// slice S (0..23) of the same: 1/24 of the RT * 6 h8 pieces -- piece index p = S * (RT * 6) / 24 ... (RT = 2: one piece every 2 slices)
template <int RT, int ACT, int S>
__device__ __forceinline__ void epi_slice(const f16v (&v)[RT][3], char* al, int w, h8& carry) {
  static_assert(RT == 2, "slicing written for RT = 2: 12 pieces of 8 values, half a piece per k-step");
  constexpr int piece = S / 2, half = S & 1;
  constexpr int r = piece / 6, t = (piece % 6) / 2, c = piece % 2;
#pragma unroll
  for (int i = 0; i < 4; ++i) carry[4 * half + i] = actf<ACT>(v[r][t][8 * c + 4 * half + i]);
  if (half == 1) *reinterpret_cast<h8*>(al + r * 32 * PITCH + ((3 * w + t) * 2 + c) * 32) = carry;
}

template <int RT, int DW, int ACT, bool EPI>
__global__ __launch_bounds__(256, 1) void seq_kernel(const h8* __restrict__ W, float* out, int layers) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n = lane & 31, h = lane >> 5;
  for (int i = tid; i < RT * 32 * PITCH / 4; i += 256) reinterpret_cast<float*>(smem)[i] = 0.001f * (i & 255);
  __syncthreads();
  char* al = smem + n * PITCH + 16 * h;
  const h8* wl = W + (size_t)w * 24 * 3 * 64 + lane;
  h8 wf[DW][3];
  f16v acc[RT][3];
  for (int L = 0; L < layers; ++L) {
#pragma unroll
    for (int d = 0; d < DW; ++d)
#pragma unroll
      for (int t = 0; t < 3; ++t) wf[d][t] = wl[(d * 3 + t) * 64];
#pragma unroll
    for (int r = 0; r < RT; ++r)
#pragma unroll
      for (int t = 0; t < 3; ++t)
#pragma unroll
        for (int k = 0; k < 16; ++k) acc[r][t][k] = 0.25f;
    h8 bf[2][RT];
#pragma unroll
    for (int r = 0; r < RT; ++r) bf[0][r] = *reinterpret_cast<const h8*>(al + r * 32 * PITCH);
#pragma unroll
    for (int s = 0; s < 24; ++s) {
      if (s + 1 < 24) {
#pragma unroll
        for (int r = 0; r < RT; ++r) bf[(s + 1) & 1][r] = *reinterpret_cast<const h8*>(al + r * 32 * PITCH + (s + 1) * 32);
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int t = 0; t < 3; ++t)
#pragma unroll
        for (int r = 0; r < RT; ++r) acc[r][t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[s % DW][t], bf[s & 1][r], acc[r][t], 0, 0, 0);
      if (s + DW < 24) {
#pragma unroll
        for (int t = 0; t < 3; ++t) wf[s % DW][t] = wl[((s + DW) * 3 + t) * 64];
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    __syncthreads();
    if (EPI) epi_all<RT, ACT>(acc, al, w);
    __syncthreads();
  }
  float s = 0.f;
  for (int r = 0; r < RT; ++r) for (int t = 0; t < 3; ++t) for (int k = 0; k < 16; ++k) s += acc[r][t][k];
  if (s == 12345.678f) out[0] = s;
}

template <int RT, int DW, int S, int ACT>
__device__ __forceinline__ void kstep(f16v (&acc)[RT][3], h8 (&wf)[DW][3], const h8* wl, const char* bl, h8 (&bf)[2][RT],
                                      const f16v (&other)[RT][3], char* ol, int w, h8& carry) {
  if (S + 1 < 24) {
#pragma unroll
    for (int r = 0; r < RT; ++r) bf[(S + 1) & 1][r] = *reinterpret_cast<const h8*>(bl + r * 32 * PITCH + (S + 1) * 32);
  }
#pragma unroll
  for (int t = 0; t < 3; ++t)
#pragma unroll
    for (int r = 0; r < RT; ++r) acc[r][t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[S % DW][t], bf[S & 1][r], acc[r][t], 0, 0, 0);
  if (S + DW < 24) {
#pragma unroll
    for (int t = 0; t < 3; ++t) wf[S % DW][t] = wl[((S + DW) * 3 + t) * 64];
  }
  epi_slice<RT, ACT, S>(other, ol, w, carry);          // VALU + one LDS store in the shadow of the 6 MFMAs just issued
  __builtin_amdgcn_sched_barrier(0);
}
template <int RT, int DW, int ACT, int S = 0>
__device__ __forceinline__ void kloop(f16v (&acc)[RT][3], h8 (&wf)[DW][3], const h8* wl, const char* bl, h8 (&bf)[2][RT],
                                      const f16v (&other)[RT][3], char* ol, int w, h8& carry) {
  if constexpr (S < 24) {
    kstep<RT, DW, S, ACT>(acc, wf, wl, bl, bf, other, ol, w, carry);
    kloop<RT, DW, ACT, S + 1>(acc, wf, wl, bl, bf, other, ol, w, carry);
  }
}

template <int DW, int ACT>
__global__ __launch_bounds__(256, 1) void pipe_kernel(const h8* __restrict__ W, float* out, int layers) {
  constexpr int RT = 2;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n = lane & 31, h = lane >> 5;
  for (int i = tid; i < 2 * RT * 32 * PITCH / 4; i += 256) reinterpret_cast<float*>(smem)[i] = 0.001f * (i & 255);
  __syncthreads();
  char* alA = smem + n * PITCH + 16 * h;
  char* alB = alA + RT * 32 * PITCH;
  const h8* wl = W + (size_t)w * 24 * 3 * 64 + lane;
  h8 wf[DW][3];
  f16v accA[RT][3], accB[RT][3];
#pragma unroll
  for (int r = 0; r < RT; ++r)
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
      for (int k = 0; k < 16; ++k) { accA[r][t][k] = 0.25f; accB[r][t][k] = 0.25f; }
  h8 carry;
  for (int L = 0; L < layers; ++L) {
    // ---- GEMM_L(A) || epilogue_{L-1}(B)
    {
      f16v acc[RT][3];
#pragma unroll
      for (int r = 0; r < RT; ++r)
#pragma unroll
        for (int t = 0; t < 3; ++t)
#pragma unroll
          for (int k = 0; k < 16; ++k) acc[r][t][k] = 0.25f;
#pragma unroll
      for (int d = 0; d < DW; ++d)
#pragma unroll
        for (int t = 0; t < 3; ++t) wf[d][t] = wl[(d * 3 + t) * 64];
      h8 bf[2][RT];
#pragma unroll
      for (int r = 0; r < RT; ++r) bf[0][r] = *reinterpret_cast<const h8*>(alA + r * 32 * PITCH);
      kloop<RT, DW, ACT>(acc, wf, wl, alA, bf, accB, alB, w, carry);
#pragma unroll
      for (int r = 0; r < RT; ++r)
#pragma unroll
        for (int t = 0; t < 3; ++t) accA[r][t] = acc[r][t];
    }
    __syncthreads();
    // ---- GEMM_L(B) || epilogue_L(A)
    {
      f16v acc[RT][3];
#pragma unroll
      for (int r = 0; r < RT; ++r)
#pragma unroll
        for (int t = 0; t < 3; ++t)
#pragma unroll
          for (int k = 0; k < 16; ++k) acc[r][t][k] = 0.25f;
#pragma unroll
      for (int d = 0; d < DW; ++d)
#pragma unroll
        for (int t = 0; t < 3; ++t) wf[d][t] = wl[(d * 3 + t) * 64];
      h8 bf[2][RT];
#pragma unroll
      for (int r = 0; r < RT; ++r) bf[0][r] = *reinterpret_cast<const h8*>(alB + r * 32 * PITCH);
      kloop<RT, DW, ACT>(acc, wf, wl, alB, bf, accA, alA, w, carry);
#pragma unroll
      for (int r = 0; r < RT; ++r)
#pragma unroll
        for (int t = 0; t < 3; ++t) accB[r][t] = acc[r][t];
    }
    __syncthreads();
  }
  float s = 0.f;
  for (int r = 0; r < RT; ++r) for (int t = 0; t < 3; ++t) for (int k = 0; k < 16; ++k) s += accA[r][t][k] + accB[r][t][k];
  if (s == 12345.678f) out[0] = s;
}

template <typename K>
void run(const char* name, K kern, int rows, int lds, const h8* W, float* out, int layers) {
  hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  for (int i = 0; i < 2; ++i) hipLaunchKernelGGL(kern, dim3(256), dim3(256), lds, 0, W, out, layers);
  hipEventRecord(a);
  const int reps = 5;
  for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(kern, dim3(256), dim3(256), lds, 0, W, out, layers);
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b); ms /= reps;
  const double per = ms * 1e3 / layers;
  printf("%-58s %3d rows/workgroup  %7.2f us per layer  %6.1f ns per row-layer  %7.1f TFLOP/s\n", name, rows, per, per * 1e3 / rows,
         2.0 * rows * 384 * 384 * 256.0 * layers / (ms * 1e-3) / 1e12);
}

int main() {
  const size_t wn = (size_t)384 * 384;
  h8* W; hipMalloc(&W, wn * 2);
  std::vector<_Float16> hw(wn); for (size_t i = 0; i < wn; ++i) hw[i] = (_Float16)(0.01f * (float)((i * 7) % 13));
  hipMemcpy(W, hw.data(), wn * 2, hipMemcpyHostToDevice);
  float* out; hipMalloc(&out, 4);
  const int L = 48;
  for (int rep = 0; rep < 2; ++rep) {
    run("seq  RT=3 DW=6  GEMM only", seq_kernel<3, 6, 1, false>, 96, 3 * 32 * PITCH, W, out, L);
    run("seq  RT=3 DW=6  + relu epilogue", seq_kernel<3, 6, 1, true>, 96, 3 * 32 * PITCH, W, out, L);
    run("seq  RT=3 DW=6  + sigmoid epilogue", seq_kernel<3, 6, 2, true>, 96, 3 * 32 * PITCH, W, out, L);
    run("seq  RT=2 DW=6  GEMM only", seq_kernel<2, 6, 1, false>, 64, 2 * 32 * PITCH, W, out, L);
    run("seq  RT=2 DW=6  + sigmoid epilogue", seq_kernel<2, 6, 2, true>, 64, 2 * 32 * PITCH, W, out, L);
    run("pipe 2 x RT=2 DW=6  relu epilogue under the other GEMM", pipe_kernel<6, 1>, 128, 4 * 32 * PITCH, W, out, L);
    run("pipe 2 x RT=2 DW=6  sigmoid epilogue under the other GEMM", pipe_kernel<6, 2>, 128, 4 * 32 * PITCH, W, out, L);
    run("pipe 2 x RT=2 DW=8  sigmoid epilogue under the other GEMM", pipe_kernel<8, 2>, 128, 4 * 32 * PITCH, W, out, L);
  }
  return 0;
}

// Dev tool: per-workgroup phase timeline of the 64->64 3x3 conv_kernel (build with -DENC_TRACE, see tools/ws_trace.hip).
#include "../dpvo_amd/csrc/encoder.hip"
#include <cstdio>
#include <vector>
int main() {
  const int H = 480, W = 640;
  const size_t wsb = dpvo_encoders_workspace_bytes(H, W);
  void *ws, *img, *fmap, *imap, *wbuf;
  hipMalloc(&ws, wsb); hipMalloc(&img, 3 * H * W * 2); hipMalloc(&fmap, (size_t)H / 4 * W / 4 * 128 * 2); hipMalloc(&imap, (size_t)H / 4 * W / 4 * 384 * 2);
  hipMalloc(&wbuf, 64 << 20); hipMemset(wbuf, 0, 64 << 20); hipMemset(img, 0, 3 * H * W * 2);
  const void* wt[44];
  for (int i = 0; i < 44; ++i) wt[i] = (char*)wbuf + (size_t)i * (1 << 20);
  for (int it = 0; it < 3; ++it) dpvo_encoders_forward(img, wt, fmap, imap, H, W, ws, wsb, nullptr);
  hipDeviceSynchronize();
  static unsigned long long h[2048][8];
  hipMemcpyFromSymbol(h, HIP_SYMBOL(g_enc_trace), sizeof(h));
  unsigned long long t0 = ~0ull, t1 = 0;
  for (int b = 0; b < 150; ++b) { if (h[b][0] && h[b][0] < t0) t0 = h[b][0]; if (h[b][5] > t1) t1 = h[b][5]; }
  printf("64->64 3x3 conv, 150 tiles per encoder: kernel span %.2f us (10 ns ticks)\n", (t1 - t0) / 100.0);
  const char* nm[] = {"entry", "stats", "halo", "mfma", "store", "end"};
  for (int b : {0, 1, 50, 100, 149}) {
    printf("block %3d:", b);
    for (int i = 0; i < 6; ++i) printf(" %s %6.2f", nm[i], (h[b][i] - t0) / 100.0);
    printf("\n");
  }
  return 0;
}

// membw3.hip -- would a sigma-hermitian coarse stencil (backward hops read from the forward arrays of the neighbour site) save HBM
// time?  Model of the level-1 coarse apply of C3 (512^2, nc = 24: 2.4 GB per matrix field): five distinct streams against three
// streams + the second and third re-read at an earlier position (one tile back = the -x neighbour's block; rows back = the -y
// neighbour's row, which another block on another XCD read a little earlier).  (diagnostic, not product)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1); } } while (0)

// block b owns elements [b*TILE, (b+1)*TILE) of every stream; stream s (< NS) is read at the block's own position, re-read r at the
// position `back[r]` elements earlier in stream 1 + r
template <int NS, int NR, bool NT>
__global__ __launch_bounds__(256) void k_model(const double2* __restrict__ m, long len, long stride, long tile, long back0, long back1, double* out) {
  double sx = 0, sy = 0;
  for (long b = blockIdx.x; b * tile < len; b += gridDim.x) {
    for (long i = b * tile + threadIdx.x; i < (b + 1) * tile && i < len; i += 256) {
      double2 r = {0, 0};
#pragma unroll
      for (int s = 0; s < NS; s++) {
        double2 v;
        if (NT) { v.x = __builtin_nontemporal_load(&m[s * stride + i].x); v.y = __builtin_nontemporal_load(&m[s * stride + i].y); }
        else v = m[s * stride + i];
        r.x += v.x; r.y += v.y;
      }
#pragma unroll
      for (int q = 0; q < NR; q++) {
        long k = i - (q == 0 ? back0 : back1);
        if (k < 0) k += len;
        const double2 v = m[(1 + q) * stride + k];
        r.x += v.x; r.y += v.y;
      }
      sx += r.x; sy += r.y;
    }
  }
  if (sx + sy == 1.2345e300) out[0] = sx;
}
template <typename F> float timeit(F f, int reps) {
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  for (int i = 0; i < 3; i++) f();
  CK(hipEventRecord(a));
  for (int i = 0; i < reps; i++) f();
  CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
  float ms; CK(hipEventElapsedTime(&ms, a, b));
  return ms / reps;
}
int main() {
  const long sites = 512L * 512, nc2 = 576;
  const long len = sites * nc2;                 // elements per matrix field (2.4 GB)
  const long tile = 5 * nc2;                    // kernel B: 5 sites per block at nc = 24
  const long row = 256 * nc2;                   // one half row of matrices (2.36 MB)
  double2* m; double* out;
  CK(hipMalloc(&m, sizeof(double2) * 5 * len)); CK(hipMalloc(&out, 8));
  CK(hipMemset(m, 1, sizeof(double2) * 5 * len));
  const int g = (int)((len + tile - 1) / tile);
  float t;
  const double all = 5.0 * len * 16.0;
#define RUN(NS, NR, NT, B0, B1, label) \
  t = timeit([&] { k_model<NS, NR, NT><<<g, 256>>>(m, len, len, tile, B0, B1, out); }, 10); \
  printf("%-72s %.3f ms  (%.0f GB/s if all five were HBM reads; HBM bytes %.2f GB)\n", label, t, all / t / 1e6, (NS * len * 16.0) / 1e9);
  RUN(5, 0, false, 0, 0, "five distinct streams")
  RUN(5, 0, true, 0, 0, "five distinct streams, non-temporal")
  RUN(3, 0, false, 0, 0, "three streams only (the lower bound)")
  RUN(3, 2, false, tile, 2 * row, "three + re-read one tile back (-x) and two half rows back (-y)")
  RUN(3, 2, true, tile, 2 * row, "same, first reads non-temporal")
  RUN(3, 2, false, tile, 3 * row, "three + re-read one tile back and three half rows back")
  RUN(3, 2, false, tile, 8 * row, "three + re-read one tile back and eight half rows back")
  RUN(3, 2, false, tile, 32 * row, "three + re-read one tile back and 32 half rows back (75 MB)")
  return 0;
}

// membw2.hip -- which feature of the stencil's access mix costs bandwidth? (diagnostic, not product)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1); } } while (0)

// NS streams of `len` elements each, `stride` elements apart; thread i reads element i of every stream.
template <int NS, bool NT, bool WR>
__global__ __launch_bounds__(256) void k_streams(const double2* __restrict__ m, long len, long stride, double2* __restrict__ y, double* out) {
  double sx = 0, sy = 0;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < len; i += (long)gridDim.x * 256) {
    double2 v[NS];
#pragma unroll
    for (int s = 0; s < NS; s++) {
      if (NT) { v[s].x = __builtin_nontemporal_load(&m[s * stride + i].x); v[s].y = __builtin_nontemporal_load(&m[s * stride + i].y); }
      else v[s] = m[s * stride + i];
    }
    double2 r = v[0];
#pragma unroll
    for (int s = 1; s < NS; s++) { r.x += v[s].x; r.y += v[s].y; }
    if (WR) { if ((i & 3) == 0) y[i >> 2] = r; }   // 1 store per 4 lanes... (stencil: 2 of 4 lanes store)
    else { sx += r.x; sy += r.y; }
  }
  if (!WR && sx + sy == 1.2345e300) out[0] = sx;
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
  const long len = 4L * 16777216;          // 1 GiB per stream (4096^2 x 4 elements)
  const long pad = 4096 + 512;             // elements of padding between streams for the "padded" case
  double2 *m, *y; double* out;
  CK(hipMalloc(&m, sizeof(double2) * (5 * (len + pad)))); CK(hipMalloc(&y, sizeof(double2) * len / 4)); CK(hipMalloc(&out, 8));
  CK(hipMemset(m, 1, sizeof(double2) * (5 * (len + pad))));
  for (int g : {8192, 65536, 1048576}) {
    float t;
#define RUN(NS, NT, WR, STRIDE, label) \
    t = timeit([&] { k_streams<NS, NT, WR><<<g, 256>>>(m, len, STRIDE, y, out); }, 10); \
    printf("grid %8d %-34s %.0f GB/s\n", g, label, (NS * len + (WR ? len / 4 : 0)) * 16.0 / t / 1e6);
    RUN(1, false, false, len, "1 stream read")
    RUN(5, false, false, len, "5 streams read, 1GiB apart")
    RUN(5, false, false, len + pad, "5 streams read, padded")
    RUN(5, true, false, len, "5 streams read nt, 1GiB apart")
    RUN(5, true, false, len + pad, "5 streams read nt, padded")
    RUN(5, false, true, len, "5 streams read + write, 1GiB")
    RUN(5, true, true, len + pad, "5 streams nt + write, padded")
  }
  return 0;
}

// membw.hip -- HBM ceiling microbenchmarks for DESIGN.md (not part of the product).
// read-only sum, copy, 5-stream read + 1 write (the stencil's stream mix), all 16 B per lane.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1); } } while (0)

__global__ __launch_bounds__(256) void k_read(const double2* __restrict__ a, long n, double* out) {
  double s = 0;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) { double2 v = a[i]; s += v.x + v.y; }
  if (s == 1.2345e300) out[0] = s;
}
__global__ __launch_bounds__(256) void k_read_nt(const double2* __restrict__ a, long n, double* out) {
  double s = 0;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    s += __builtin_nontemporal_load(&a[i].x) + __builtin_nontemporal_load(&a[i].y);
  }
  if (s == 1.2345e300) out[0] = s;
}
__global__ __launch_bounds__(256) void k_copy(const double2* __restrict__ a, double2* __restrict__ b, long n) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) b[i] = a[i];
}
// stencil mix: 11 x 16B reads (5 matrix streams of 4n... modelled as 10 equal streams + 1) : here 5 streams read, 1 stream (1/10 size) read, 1 write
__global__ __launch_bounds__(256) void k_mix(const double2* __restrict__ m, long n, const double2* __restrict__ x, double2* __restrict__ y) {
  // m: 5 arrays of 4n elements; x,y: 2n elements ; thread t handles element t of 4n, writes y when (t&1)==0
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < 4 * n; i += (long)gridDim.x * 256) {
    double2 a0 = m[i], a1 = m[4 * n + i], a2 = m[8 * n + i], a3 = m[12 * n + i], a4 = m[16 * n + i];
    double2 xv = x[i >> 1];
    double2 r;
    r.x = a0.x * xv.x + a1.x * xv.y + a2.x + a3.x + a4.x;
    r.y = a0.y * xv.x + a1.y * xv.y + a2.y + a3.y + a4.y;
    if ((i & 1) == 0) y[i >> 1] = r;
  }
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
int main(int argc, char** argv) {
  long n = 16777216;  // sites (4096^2)
  double2 *m, *x, *y; double* out;
  CK(hipMalloc(&m, sizeof(double2) * 20 * n)); CK(hipMalloc(&x, sizeof(double2) * 2 * n)); CK(hipMalloc(&y, sizeof(double2) * 2 * n)); CK(hipMalloc(&out, 8));
  CK(hipMemset(m, 1, sizeof(double2) * 20 * n)); CK(hipMemset(x, 1, sizeof(double2) * 2 * n));
  const long N = 20 * n;
  for (int g : {2048, 4096, 8192, 16384, 65536, 262144}) {
    float t = timeit([&] { k_read<<<g, 256>>>(m, N, out); }, 20);
    float tn = timeit([&] { k_read_nt<<<g, 256>>>(m, N, out); }, 20);
    float tc = timeit([&] { k_copy<<<g, 256>>>(m, m + 10 * n, 10 * n); }, 20);
    float tm = timeit([&] { k_mix<<<g, 256>>>(m, n, x, y); }, 20);
    printf("grid %7d: read %.1f GB/s  read_nt %.1f GB/s  copy %.1f GB/s (r+w)  stencil-mix %.1f GB/s\n", g, N * 16 / t / 1e6, N * 16 / tn / 1e6,
           2.0 * 10 * n * 16 / tc / 1e6, (20.0 * n + 4.0 * n) * 16 / tm / 1e6);
  }
  return 0;
}

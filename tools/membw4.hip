// membw4.hip -- which cache-policy bits should a streaming matrix load carry on gfx950?  Five 2.4 GB streams (the level-1 coarse operator's
// bytes), one contiguous tile per block as in membw3, global_load_dwordx4 with every combination of sc0 / sc1 / nt.  (diagnostic, not product)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1); } } while (0)
typedef float v4f __attribute__((ext_vector_type(4)));

template <int POL> __device__ __forceinline__ v4f ldp(const v4f* p) {
  v4f r;
  if (POL == 0) asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(r) : "v"(p) : "memory");
  if (POL == 1) asm volatile("global_load_dwordx4 %0, %1, off sc0" : "=v"(r) : "v"(p) : "memory");
  if (POL == 2) asm volatile("global_load_dwordx4 %0, %1, off nt" : "=v"(r) : "v"(p) : "memory");
  if (POL == 3) asm volatile("global_load_dwordx4 %0, %1, off sc0 nt" : "=v"(r) : "v"(p) : "memory");
  if (POL == 4) asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(r) : "v"(p) : "memory");
  if (POL == 5) asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1" : "=v"(r) : "v"(p) : "memory");
  if (POL == 6) asm volatile("global_load_dwordx4 %0, %1, off sc1 nt" : "=v"(r) : "v"(p) : "memory");
  if (POL == 7) asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1 nt" : "=v"(r) : "v"(p) : "memory");
  return r;
}

template <int POL>
__global__ __launch_bounds__(256) void k_model(const v4f* __restrict__ m, long len, long tile, float* out) {
  float s = 0;
  for (long b = blockIdx.x; b * tile < len; b += gridDim.x) {
    for (long i = b * tile + threadIdx.x; i < (b + 1) * tile && i < len; i += 256) {
      v4f v0 = ldp<POL>(m + i), v1 = ldp<POL>(m + len + i), v2 = ldp<POL>(m + 2 * len + i), v3 = ldp<POL>(m + 3 * len + i), v4 = ldp<POL>(m + 4 * len + i);
      asm volatile("s_waitcnt vmcnt(0)" : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+v"(v4) :: "memory");
      s += v0.x + v1.y + v2.z + v3.w + v4.x;
    }
  }
  if (s == 1.2345e30f) out[0] = s;
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
  const long len = 512L * 512 * 576, tile = 5 * 576;
  v4f* m; float* out;
  CK(hipMalloc(&m, sizeof(v4f) * 5 * len)); CK(hipMalloc(&out, 4));
  CK(hipMemset(m, 1, sizeof(v4f) * 5 * len));
  const int g = (int)((len + tile - 1) / tile);
  const char* names[8] = {"(none)", "sc0", "nt", "sc0 nt", "sc1", "sc0 sc1", "sc1 nt", "sc0 sc1 nt"};
  float t;
#define RUN(P) t = timeit([&] { k_model<P><<<g, 256>>>(m, len, tile, out); }, 10); printf("policy %-12s %.3f ms  %.0f GB/s\n", names[P], t, 5.0 * len * 16.0 / t / 1e6);
  for (int rep = 0; rep < 2; rep++) { RUN(0) RUN(1) RUN(2) RUN(3) RUN(4) RUN(5) RUN(6) RUN(7) }
  return 0;
}

// Back-to-back v_mfma_f64_16x16x4_f64 issue rate on gfx950: one wave per SIMD, NACC independent accumulators.
// hipcc -O3 --offload-arch=gfx950 -o mfma_f64_rate mfma_f64_rate.hip && ./mfma_f64_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef double v4d __attribute__((ext_vector_type(4)));
template <int NACC>
__global__ __launch_bounds__(256) void k(double* out, int iters, double a0, double b0) {
  v4d acc[NACC];
  for (int i = 0; i < NACC; i++) acc[i] = (v4d){0, 0, 0, 0};
  double a = a0 + threadIdx.x, b = b0 - threadIdx.x;
  long long t0 = clock64();
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int i = 0; i < NACC; i++) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
  }
  long long t1 = clock64();
  double s = 0;
  for (int i = 0; i < NACC; i++) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = (double)(t1 - t0);
}
template <int NACC>
void run(int blocks_per_cu) {
  int iters = 20000;
  double* d;
  int nb = 256 * blocks_per_cu;
  hipMalloc(&d, sizeof(double) * 256 * nb);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  k<NACC><<<nb, 256>>>(d, 10, 1.0, 2.0);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  k<NACC><<<nb, 256>>>(d, iters, 1.0, 2.0);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  double clk;
  hipMemcpy(&clk, d, 8, hipMemcpyDeviceToHost);
  double n_mfma = (double)iters * NACC * 4.0 * nb;   // wave-level instructions
  double tf = n_mfma * 2048.0 / (ms * 1e-3) / 1e12;
  printf("NACC %d, %d blocks/CU: %.3f ms, %.1f TFLOP/s f64, %.1f clock64 ticks per MFMA per wave (100 MHz counter)\n", NACC, blocks_per_cu, ms, tf, clk / (iters * (double)NACC));
  hipFree(d);
}
int main() {
  run<1>(1); run<4>(1); run<8>(1); run<4>(2); run<8>(2);
  return 0;
}

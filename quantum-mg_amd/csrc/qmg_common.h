// qmg_common.h -- shared device helpers for the gfx950 kernels of libqmg_hip.so.
#ifndef QMG_COMMON_H
#define QMG_COMMON_H

#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include "../../include/qmg_hip.h"

namespace qmg {

typedef double2 cplx;   // (x = re, y = im): one 16-byte global_load_dwordx4 / global_store_dwordx4

constexpr int WAVE = 64;          // CDNA4 wavefront
constexpr int BLOCK = 256;        // 4 waves = one per SIMD

// last HIP error text, for qmg_last_hip_error()
void set_hip_error(hipError_t e, const char* where);

#define QMG_HIP_CHECK(expr)                                  \
  do {                                                       \
    hipError_t _e = (expr);                                  \
    if (_e != hipSuccess) {                                  \
      ::qmg::set_hip_error(_e, #expr);                       \
      return QMG_ERR_HIP;                                    \
    }                                                        \
  } while (0)

#define QMG_LAUNCH_CHECK()                                   \
  do {                                                       \
    hipError_t _e = hipGetLastError();                       \
    if (_e != hipSuccess) {                                  \
      ::qmg::set_hip_error(_e, "kernel launch");             \
      return QMG_ERR_HIP;                                    \
    }                                                        \
  } while (0)

inline hipStream_t as_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }

inline bool valid_lattice(int Lx, int Ly) { return Lx >= 2 && Ly >= 2 && !(Lx & 1) && !(Ly & 1); }

// ---------------- complex arithmetic (explicit FMAs; 8 flop per MAC) ----------------
__device__ __forceinline__ cplx cmake(double re, double im) { return make_double2(re, im); }
__device__ __forceinline__ cplx cadd(cplx a, cplx b) { return make_double2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ cplx csub(cplx a, cplx b) { return make_double2(a.x - b.x, a.y - b.y); }
__device__ __forceinline__ cplx cconj(cplx a) { return make_double2(a.x, -a.y); }
__device__ __forceinline__ cplx cmul(cplx a, cplx b) {
  return make_double2(fma(a.x, b.x, -a.y * b.y), fma(a.x, b.y, a.y * b.x));
}
// acc += a*b
__device__ __forceinline__ void cmac(cplx& acc, cplx a, cplx b) {
  acc.x = fma(a.x, b.x, acc.x);
  acc.x = fma(-a.y, b.y, acc.x);
  acc.y = fma(a.x, b.y, acc.y);
  acc.y = fma(a.y, b.x, acc.y);
}
// acc += conj(a)*b
__device__ __forceinline__ void cmac_conj(cplx& acc, cplx a, cplx b) {
  acc.x = fma(a.x, b.x, acc.x);
  acc.x = fma(a.y, b.y, acc.x);
  acc.y = fma(a.x, b.y, acc.y);
  acc.y = fma(-a.y, b.x, acc.y);
}

// ---------------- cross-lane moves on doubles (DPP; no LDS traffic) ----------------
template <int CTRL>
__device__ __forceinline__ double dpp_move(double v) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_mov_dpp(lo, CTRL, 0xF, 0xF, true);
  hi = __builtin_amdgcn_mov_dpp(hi, CTRL, 0xF, 0xF, true);
  return __hiloint2double(hi, lo);
}
// quad_perm:[1,0,3,2] -> lane ^ 1 ; quad_perm:[2,3,0,1] -> lane ^ 2
__device__ __forceinline__ double lane_xor1(double v) { return dpp_move<0xB1>(v); }
__device__ __forceinline__ double lane_xor2(double v) { return dpp_move<0x4E>(v); }

// Full 64-lane sum; every lane gets the total. (ds_swizzle/bpermute-free: row_shr + bcast is
// order-dependent, so use the butterfly via __shfl_xor for the cross-row steps.)
__device__ __forceinline__ double wave_sum(double v) {
  v += lane_xor1(v);
  v += lane_xor2(v);
  v += __shfl_xor(v, 4);
  v += __shfl_xor(v, 8);
  v += __shfl_xor(v, 16);
  v += __shfl_xor(v, 32);
  return v;
}
__device__ __forceinline__ double wave_max(double v) {
  v = fmax(v, lane_xor1(v));
  v = fmax(v, lane_xor2(v));
  v = fmax(v, __shfl_xor(v, 4));
  v = fmax(v, __shfl_xor(v, 8));
  v = fmax(v, __shfl_xor(v, 16));
  v = fmax(v, __shfl_xor(v, 32));
  return v;
}

// Memory-bound 1-D launches.  One 16-byte element per thread up to 2^18 blocks, grid-stride beyond: on this part a
// streaming copy reaches 6.2 TB/s at 262 144 blocks but only 5.4 TB/s at 8 192 (profiles/r01_membw_ceiling.txt).
inline unsigned grid_1d(size_t work_items, int per_block = BLOCK) {
  size_t b = (work_items + per_block - 1) / per_block;
  const size_t cap = 262144u;
  if (b > cap) b = cap;
  if (b == 0) b = 1;
  return (unsigned)b;
}

}  // namespace qmg

#endif

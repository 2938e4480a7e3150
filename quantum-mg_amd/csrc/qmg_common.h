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

struct SlabHalo { const void* lo; const void* hi; long stride; int rows; };   // y-slab halos (qmg_stencil_apply_slab)
// qmg_site.hip: nc = 2 apply; storage 0 = complex<half> matrices + complex<float> vectors, 1 = complex<float>, 2 = complex<double>
int site_kernel_apply(int storage, const qmg_stencil_desc* d, void* lhs, const void* rhs, unsigned pieces, int n, long vec_stride,
                      const unsigned char* ridx, hipStream_t st, bool only_where_faster, const struct SlabHalo* slab);
int generic_slab_apply(const qmg_stencil_desc* d, void* lhs, const void* rhs, unsigned pieces, int n, long vec_stride, const unsigned char* ridx,
                       hipStream_t st, const SlabHalo* slab, int mat32, int vec32);   // qmg_stencil.hip: kernel B with halos (any nc; f32: complex<float> matrices and vectors)
constexpr int SITE_DECLINED = 1000;   // not an error: the caller's own kernel is the better one for this launch

// qmg_comm.hip: reductions of y-slab vectors are summed over the ranks (qmg_comm_set_distributed_reductions)
bool dist_reductions_on();
int dist_allreduce(double* buf_dev, int n, bool op_max, hipStream_t st);

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

// ---------------- storage precision (qmg_dtype) ----------------
// Vectors and matrices are STORED as complex<double> (QMG_C64) or complex<float> (QMG_C32).  Except in the fine-stencil
// kernel A (which computes in the storage type), arithmetic and every reduction accumulate in fp64 registers: an element
// is widened when loaded and rounded once when stored, so an fp32 kernel moves half the bytes of its fp64 twin and
// differs from it by one rounding per stored element.
template <typename T> struct CStore;
template <> struct CStore<double> { typedef double2 type; };
template <> struct CStore<float> { typedef float2 type; };

template <typename T>
__device__ __forceinline__ cplx ldc(const void* base, long i) {
  if (sizeof(T) == 8) return reinterpret_cast<const cplx*>(base)[i];
  const float2 v = reinterpret_cast<const float2*>(base)[i];
  return make_double2((double)v.x, (double)v.y);
}
// read-once stream: bypass-on-evict hint
template <typename T>
__device__ __forceinline__ cplx ldc_nt(const void* base, long i) {
  if (sizeof(T) == 8) {
    const cplx* p = reinterpret_cast<const cplx*>(base) + i;
    cplx v;
    v.x = __builtin_nontemporal_load(&p->x);
    v.y = __builtin_nontemporal_load(&p->y);
    return v;
  }
  const long long raw = __builtin_nontemporal_load(reinterpret_cast<const long long*>(base) + i);
  return make_double2((double)__int_as_float((int)(raw & 0xFFFFFFFFll)), (double)__int_as_float((int)(raw >> 32)));
}
// ---- loads that keep the STORAGE form.  A conversion, mask or shift placed right behind a load makes the compiler wait for that load -- and, the
// counter being in-order, for every load issued before it -- on the spot: a loop of `value = widen(load)` runs one memory latency per element
// (seen in the ISA of every complex<float> kernel written that way: load, s_waitcnt vmcnt(0), convert, load, ...).  Staging code therefore
// loads RAW values first (all of them), and widens them where they are consumed.
template <typename T> struct RawC { typedef cplx type; };            // complex<double>: the value itself
template <> struct RawC<float> { typedef long long type; };          // the raw bits of a complex<float>
template <typename T> __device__ __forceinline__ typename RawC<T>::type ld_raw(const void* base, long i) {
  if constexpr (sizeof(T) == 8) return reinterpret_cast<const cplx*>(base)[i];
  else return reinterpret_cast<const long long*>(base)[i];
}
template <typename T> __device__ __forceinline__ typename RawC<T>::type ld_raw_nt(const void* base, long i) {
  if constexpr (sizeof(T) == 8) {
    const cplx* p = reinterpret_cast<const cplx*>(base) + i;
    cplx v;
    v.x = __builtin_nontemporal_load(&p->x);
    v.y = __builtin_nontemporal_load(&p->y);
    return v;
  } else return __builtin_nontemporal_load(reinterpret_cast<const long long*>(base) + i);
}
template <typename T> __device__ __forceinline__ typename RawC<T>::type zero_rawc() {
  if constexpr (sizeof(T) == 8) return make_double2(0.0, 0.0);
  else return 0ll;
}
template <typename T> __device__ __forceinline__ cplx widen_rawc(typename RawC<T>::type r) {
  if constexpr (sizeof(T) == 8) return r;
  else return make_double2((double)__int_as_float((int)(r & 0xFFFFFFFFll)), (double)__int_as_float((int)(r >> 32)));
}
// W consecutive elements as ONE 16-byte raw value ((double, 1) and (float, 2)), or (float, 1) as 8 bytes
template <typename T, int W> struct RawP { typedef typename RawC<T>::type type; };
template <> struct RawP<float, 2> { typedef float type __attribute__((ext_vector_type(4))); };
template <typename T, int W> __device__ __forceinline__ typename RawP<T, W>::type ld_rawp(const void* base, long ipack) {
  if constexpr (sizeof(T) == 4 && W == 2) return reinterpret_cast<const typename RawP<float, 2>::type*>(base)[ipack];
  else return ld_raw<T>(base, ipack);
}
template <typename T, int W> __device__ __forceinline__ typename RawP<T, W>::type ld_rawp_nt(const void* base, long ipack) {
  if constexpr (sizeof(T) == 4 && W == 2) return __builtin_nontemporal_load(reinterpret_cast<const typename RawP<float, 2>::type*>(base) + ipack);
  else return ld_raw_nt<T>(base, ipack);
}
template <typename T, int W> __device__ __forceinline__ void widen_rawp(typename RawP<T, W>::type r, cplx (&v)[W]) {
  if constexpr (sizeof(T) == 4 && W == 2) { v[0] = make_double2((double)r.x, (double)r.y); v[W - 1] = make_double2((double)r.z, (double)r.w); }
  else v[0] = widen_rawc<T>(r);
}
template <typename T, int W> __device__ __forceinline__ typename RawP<T, W>::type zero_rawp() {
  if constexpr (sizeof(T) == 4 && W == 2) { typename RawP<float, 2>::type z = {0.0f, 0.0f, 0.0f, 0.0f}; return z; }
  else return zero_rawc<T>();
}
template <typename T>
__device__ __forceinline__ void stc(void* base, long i, cplx v) {
  if (sizeof(T) == 8) reinterpret_cast<cplx*>(base)[i] = v;
  else reinterpret_cast<float2*>(base)[i] = make_float2((float)v.x, (float)v.y);
}
// W consecutive elements starting at element i*W: one 16-byte access per lane for (double, W = 1) and (float, W = 2).
// The (float, 2) form needs base 16-byte aligned (the launchers check and fall back to W = 1).
template <typename T, int W>
__device__ __forceinline__ void ldc_pack(const void* base, long ipack, cplx (&v)[W]) {
  if (sizeof(T) == 4 && W == 2) {
    const float4 r = reinterpret_cast<const float4*>(base)[ipack];
    v[0] = make_double2((double)r.x, (double)r.y);
    v[W - 1] = make_double2((double)r.z, (double)r.w);
  } else {
#pragma unroll
    for (int w = 0; w < W; w++) v[w] = ldc<T>(base, ipack * W + w);
  }
}
template <typename T, int W>
__device__ __forceinline__ void ldc_pack_nt(const void* base, long ipack, cplx (&v)[W]) {   // read-once streams
  if (sizeof(T) == 4 && W == 2) {
    typedef float f4 __attribute__((ext_vector_type(4)));
    const f4 r = __builtin_nontemporal_load(reinterpret_cast<const f4*>(base) + ipack);
    v[0] = make_double2((double)r.x, (double)r.y);
    v[W - 1] = make_double2((double)r.z, (double)r.w);
  } else {
#pragma unroll
    for (int w = 0; w < W; w++) v[w] = ldc_nt<T>(base, ipack * W + w);
  }
}
template <typename T, int W>
__device__ __forceinline__ void stc_pack(void* base, long ipack, const cplx (&v)[W]) {
  if (sizeof(T) == 4 && W == 2) {
    reinterpret_cast<float4*>(base)[ipack] = make_float4((float)v[0].x, (float)v[0].y, (float)v[W - 1].x, (float)v[W - 1].y);
  } else {
#pragma unroll
    for (int w = 0; w < W; w++) stc<T>(base, ipack * W + w, v[w]);
  }
}
inline size_t dtype_size(int dtype) { return dtype == QMG_C32 ? 8 : 16; }
inline bool valid_dtype(int dtype) { return dtype == QMG_C64 || dtype == QMG_C32; }
inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

// ---------------- lock-step batches: the active systems of a call ----------------
constexpr int BATCH_MAX = 16;
struct BatchIdx { int n; int nt; unsigned char id[BATCH_MAX]; };   // nt: the batch is large enough for non-temporal reads (qmg_batch.hip)
inline BatchIdx expand_mask(unsigned mask, int nrhs) {
  BatchIdx b;
  b.n = 0;
  b.nt = 0;
  for (int k = 0; k < nrhs && k < BATCH_MAX; k++)
    if ((mask >> k) & 1u) b.id[b.n++] = (unsigned char)k;
  for (int k = b.n; k < BATCH_MAX; k++) b.id[k] = 0;
  return b;
}

// ---------------- apply epilogue (qmg_stencil_apply_epi_t, qmg_wilson_*_direct_epi): what a stencil kernel does with a finished site value ----------------
//   out = other_scale * other + acc_scale * acc      (other == nullptr: out = acc)      -- residuals b - A x, the Schur complement's r_e - D_eo t
//   dotv != nullptr: per-wavefront partial sums of conj(dotv) out (re, im) and |out|^2 go to `part`, [system slot][npart][4]   -- MR's <p,r>, <p,p>
// `other` and `dotv` are vectors with the layout, precision, stride and system numbering of lhs; only the parities the launch processes are
// touched.  Needs overwrite semantics (QMG_P_ZERO on the processed parities).  The dots are taken of the value AS STORED (rounded to the
// storage precision first), so they are the dots a separate pass over the stored vector would form.
struct Epilogue {
  const void* other;
  const void* dotv;
  double other_scale, acc_scale;
  double* part;
  long npart;        // partial slots per system (wavefronts of the launch)
  int on;
};
inline Epilogue no_epilogue() { Epilogue e; e.other = nullptr; e.dotv = nullptr; e.other_scale = 0.0; e.acc_scale = 1.0; e.part = nullptr; e.npart = 0; e.on = 0; return e; }
// qmg_batch.hip: the calling thread's MR slot.  begin: a partial buffer of nsys * npart * 4 doubles (grown on demand; nullptr on failure);
// finish: second stage -- the partials of the n systems `ids` summed in index order into the slot (conjugated to <p,r>), summed over ranks
// under distributed reductions.
double* mr_epilogue_begin(int nsys, long npart);
int mr_epilogue_finish(const unsigned char* ids, int n, long npart, hipStream_t st);

extern int g_reduce_spin;     // qmg_batch.hip; "reduce_spin"
extern int g_malloc_poison;   // qmg_runtime.hip; "malloc_poison"
// qmg_shutdown: the calling thread's reduction / norm workspaces (qmg_blas.hip, qmg_batch.hip, qmg_stencil.hip)
void release_blas_workspace();
void release_batch_workspace();
void release_stencil_workspace();
extern int g_setup_fused; // qmg_setup.hip; "setup_fused"
extern int g_wilson_pair;
extern long g_blas_nt_bytes;   // qmg_blas.hip; "blas_nt_mb"   // qmg_wilson.hip; "wilson_pair"
extern int g_xfer_mfma;   // qmg_transfer_mfma.hip; "xfer_mfma"
extern int g_xfer_pack;   // qmg_transfer.hip; "xfer_pack"
extern int g_xfer_tile;   // qmg_transfer.hip; set through qmg_set_tuning("xfer_tile", v)
extern int g_site_block, g_site_gy, g_site_generic;   // qmg_site.hip; "site_block", "site_gy", "site_generic"

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

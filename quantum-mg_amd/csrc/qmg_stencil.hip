// qmg_stencil.hip -- fused even-odd stencil apply for gfx950 (MI355X).
//
// Replaces the reference's un-fused pass structure
//     apply_M = clover sweep + 8 x {cshift copy, cMATxpy sweep} + 2 x caxpy      (stencil_2d.h:912-936)
// (~27 vector passes + 5 matrix passes) with ONE launch that reads every stencil
// matrix once, the right-hand side once (from HBM; neighbours come back out of L2) and writes
// the result once: 5 nc^2 c + 2 nc c bytes per site, the algorithmic minimum.
//
// Addressing (derived from lattice.h:75-81,204 and the loops of cshift_2d.h:60-119,149-210):
// an output site of parity p on row y at half-row column j has x = 2j + s, s = (y+p)&1, and its
// four neighbours live in the OPPOSITE parity half at
//     +x: (y, j+s)   -x: (y, j+s-1)   +y: (y+1, j)   -y: (y-1, j)        (periodic)
// The matrix multiplying a neighbour is stored at the OUTPUT site (stencil_2d.h:718-732), so a
// lane streams five arrays at one common offset.
//
// Kernel A (nc = 1, 2, 4): "element per lane".  nc^2 adjacent lanes own one site; lane (r,c)
// loads element M[r][c] of each of the five matrices, so every matrix load instruction of a
// wavefront is one fully coalesced 1 KiB segment (16 B per lane) -- the AoS (c1,c2)-fastest
// layout of the reference is already lane-contiguous and is NOT repacked.  The sum over c is a
// DPP quad-permute (no LDS).  Rows of both parities are interleaved in block order, so the
// even- and odd-output rows that share right-hand-side data run back to back on the same XCD
// (blocks per row is a multiple of 8 for power-of-two lattices) and the second use hits L2.
//
// Kernel B (any nc, used for the Galerkin coarse operators): see below.

#include <string.h>
#include <type_traits>

#include "qmg_common.h"

namespace qmg {

struct StencilArgs {
  const cplx* clover;
  const cplx* hopping;
  void* lhs;         // vectors: complex<double>, or complex<float> when vec32
  const void* rhs;
  int hr;            // Lx / 2: sites per half row
  int Ly;
  long half_vol;     // sites per parity
  long size_cm;      // complex elements per matrix field (both parities)
  unsigned pieces;
  int nrhs;
  long vec_stride;   // complex elements between right-hand sides
  int par_first;     // first parity processed
  int par_count;     // 1 or 2 (2: rows interleaved even/odd)
  int nrows;         // Ly * par_count
  double shift[2], eo_shift[2], dof_shift[2];
  unsigned char ridx[16];   // masked batches (qmg_stencil_apply_batch): right-hand side processed as column k; else unused
  int use_idx;       // 0: column k is right-hand side k
  int mat32;         // 1: clover / hopping point to complex<float> arrays (kernels B and C: qmg_stencil_apply_mat32, qmg_stencil_apply_t)
  int mat16;         // 1 (with mat32 = 1): they point to complex<half> arrays; the tile is widened to complex<float> on its way into LDS (kernel B32: qmg_stencil_apply_mat16)
  int vec32;         // 1: lhs / rhs are complex<float> (qmg_stencil_apply_t with QMG_C32: matrices AND vectors fp32)
  // y-slab of a larger lattice (kernel B only; qmg_stencil_apply_slab): rows -1 / Ly of the right-hand side come from these
  // buffers ([system][parity][hr][nc] complex, halo_stride elements between systems) instead of the periodic wrap
  const void* halo_lo;
  const void* halo_hi;
  long halo_stride;
  // fused |lhs_k|^2 (kernel A2 with NORM, qmg_stencil_apply_norm2): one partial per (row group, block, wavefront, system)
  double* norm_part;
  // apply epilogue (kernels B / B32, one system per launch): out = other_scale other + acc_scale acc, MR dots of out (qmg_common.h)
  Epilogue epi;
#ifdef QMG_DIAGNOSTICS
  int ablate;        // tools-only build (make DIAG=1; tools/variants.py): 1 = neighbours := own site, 2 = no store, 4 = no rhs loads
#endif
};

// Ablation switches exist only in the tools build (-DQMG_DIAGNOSTICS, `make DIAG=1`, never shipped): in libqmg_hip.so the
// test is the constant 0 and every ablated path is dead code.
#ifdef QMG_DIAGNOSTICS
#define QMG_ABLATE(a, bits) ((a).ablate & (bits))
#else
#define QMG_ABLATE(a, bits) 0
#endif

// The system a launch's k-th right-hand side belongs to (masked batches process a subset: a.ridx), WITHOUT touching memory: a.ridx[k] with a
// run-time k -- divergent or uniform -- is a vector load from the kernel-argument segment, and the `s_waitcnt vmcnt(0)` in front of its use also
// waits for every load issued before it: in kernels B / B32 that was the next piece's matrix prefetch, issued a few instructions earlier (the
// wavefront then sat out the whole latency before it computed on the current piece), in kernel C one more memory latency in front of every
// piece.  The sixteen bytes are four scalar registers; a lane picks its byte with selects and a shift.
__device__ __forceinline__ int system_index(const StencilArgs& a, int k) {
  unsigned long long w[2];
  __builtin_memcpy(w, a.ridx, 16);
  // one select and one shift (for a uniform k: scalar instructions).  A chain of selects per bit came out as a chain of scalar BRANCHES inside kernel
  // A2's next-system prefetch, whose load clauses they cut: 8 systems 0.93 -> 1.01 ms.
  const unsigned long long ww = (k & 8) ? w[1] : w[0];
  const int idx = (int)((ww >> (8 * (k & 7))) & 0xffull);
  return a.use_idx ? idx : k;
}
__device__ __forceinline__ long rhs_offset(const StencilArgs& a, int k) { return (long)system_index(a, k) * a.vec_stride; }
// Kernel A2 (k_stencil_pair) keeps the CONDITIONAL byte load: it is executed for masked batches only, and with it the compiler's schedule of the
// next-system prefetch is the faster one -- same box, 4096^2 staggered, 8 systems: 0.93 ms against 1.01 ms with system_index, whose code is free of
// the load but makes the compiler spread the waits of the two systems' requests differently; an explicit drain in front of the prefetch did not
// bring the 0.93 back (tools/apply_norm_ab.py, gpurun_out/ab_*.txt).  Measured, not understood.
__device__ __forceinline__ long rhs_offset_a(const StencilArgs& a, int k) { return (long)(a.use_idx ? (int)a.ridx[k] : k) * a.vec_stride; }

// vector element i of a complex<double> (V32 = false) or complex<float> (V32 = true) array, in fp64 registers
template <bool V32> __device__ __forceinline__ cplx ldv(const void* base, long i) { return V32 ? ldc<float>(base, i) : ldc<double>(base, i); }
template <bool V32> __device__ __forceinline__ void stv(void* base, long i, cplx v) { if (V32) stc<float>(base, i, v); else stc<double>(base, i, v); }
// A vector element in its STORAGE form (V32: the raw bits of a complex<float> in a double) and its widening.  Staging registers hold the raw
// form: a conversion right behind the load makes the compiler wait for that load -- and for everything issued before it -- on the spot.
template <bool V32> struct XRaw { typedef cplx type; };
template <> struct XRaw<true> { typedef double type; };
template <bool V32> __device__ __forceinline__ typename XRaw<V32>::type ldv_raw(const void* base, long i) {
  if constexpr (V32) return reinterpret_cast<const double*>(base)[i];
  else return reinterpret_cast<const cplx*>(base)[i];
}
template <bool V32> __device__ __forceinline__ cplx widen_raw(typename XRaw<V32>::type v) {
  if constexpr (V32) { struct F2 { float x, y; }; const F2 f = __builtin_bit_cast(F2, v); return cmake((double)f.x, (double)f.y); }
  else return v;
}
template <bool V32> __device__ __forceinline__ typename XRaw<V32>::type zero_raw() {
  if constexpr (V32) return 0.0;
  else return cmake(0.0, 0.0);
}

// the epilogue of one output element (qmg_common.h: Epilogue): returns the value to store, accumulates the MR dots of the value AS STORED
// ov / r: the element's `other` / `dotv` values, loaded by the caller at the START of the row (a load issued here, after the tile loop,
// would add a full memory latency to every block)
template <bool V32>
__device__ __forceinline__ cplx epilogue_value(const Epilogue& e, cplx ov, cplx r, cplx t, double (&d)[3]) {
  if (e.other) t = cmake(fma(e.other_scale, ov.x, e.acc_scale * t.x), fma(e.other_scale, ov.y, e.acc_scale * t.y));
  else if (e.acc_scale != 1.0) t = cmake(e.acc_scale * t.x, e.acc_scale * t.y);
  if (e.dotv) {
    const cplx sv = V32 ? cmake((double)(float)t.x, (double)(float)t.y) : t;
    d[0] = fma(r.x, sv.x, d[0]); d[0] = fma(r.y, sv.y, d[0]);      // conj(r) out
    d[1] = fma(r.x, sv.y, d[1]); d[1] = fma(-r.y, sv.x, d[1]);
    d[2] = fma(sv.x, sv.x, d[2]); d[2] = fma(sv.y, sv.y, d[2]);
  }
  return t;
}
// end of a kernel with an epilogue: one partial per wavefront of the launch, [slot][4] (system slot 0); every lane of the block calls it
// (the launchers cap grid.y for these launches, so that the one-block second stage sums a few thousand partials, not one per row)
__device__ __forceinline__ void epilogue_store_partials(const Epilogue& e, double (&d)[3]) {
  const double s0 = wave_sum(d[0]), s1 = wave_sum(d[1]), s2 = wave_sum(d[2]);
  if ((threadIdx.x & (WAVE - 1)) == 0) {
    const long w = ((long)blockIdx.y * gridDim.x + blockIdx.x) * (BLOCK / WAVE) + threadIdx.x / WAVE;
    double* p = e.part + w * 4;
    p[0] = s0; p[1] = s1; p[2] = s2; p[3] = 0.0;
  }
}

template <bool NT>
__device__ __forceinline__ cplx ld(const cplx* p) {
  if (NT) {
    cplx v;
    v.x = __builtin_nontemporal_load(&p->x);
    v.y = __builtin_nontemporal_load(&p->y);
    return v;
  }
  return *p;
}

// matrix element i of a complex<double> (M32 = false) or complex<float> (M32 = true) array, widened to fp64
template <bool M32, bool NT>
__device__ __forceinline__ cplx ldm(const cplx* base, long i) {
  if (M32) {
    const float2* p = reinterpret_cast<const float2*>(base) + i;
    float2 v;
    if (NT) {
      const long long raw = __builtin_nontemporal_load(reinterpret_cast<const long long*>(p));
      v.x = __int_as_float((int)(raw & 0xFFFFFFFFll));
      v.y = __int_as_float((int)(raw >> 32));
    } else v = *p;
    return make_double2((double)v.x, (double)v.y);
  }
  return ld<NT>(base + i);
}

// a matrix element in its STORAGE form (M32: the raw 8 bytes of a complex<float>) -- staging registers hold this, the widening happens where the
// element is parked (qmg_common.h: a conversion behind each load serialises the loads)
template <bool M32> struct MRaw { typedef cplx type; };
template <> struct MRaw<true> { typedef long long type; };
template <bool M32, bool NT> __device__ __forceinline__ typename MRaw<M32>::type ldm_raw(const cplx* base, long i) {
  if constexpr (M32) {
    const long long* p = reinterpret_cast<const long long*>(base) + i;
    return NT ? __builtin_nontemporal_load(p) : *p;
  } else return ld<NT>(base + i);
}
template <bool M32> __device__ __forceinline__ cplx widen_mraw(typename MRaw<M32>::type r) {
  if constexpr (M32) return make_double2((double)__int_as_float((int)(r & 0xFFFFFFFFll)), (double)__int_as_float((int)(r >> 32)));
  else return r;
}
template <bool M32> __device__ __forceinline__ typename MRaw<M32>::type zero_mraw() {
  if constexpr (M32) return 0ll;
  else return make_double2(0.0, 0.0);
}
// two consecutive complex<float> matrix elements (16 B, element index i even) widened to fp64
template <bool NT>
__device__ __forceinline__ void ldm32_pair(const cplx* base, long i, cplx& v0, cplx& v1) {
  const double* p = reinterpret_cast<const double*>(reinterpret_cast<const float2*>(base) + i);   // 16-B aligned for even i
  long long r0, r1;
  if (NT) {
    r0 = __builtin_nontemporal_load(reinterpret_cast<const long long*>(p));
    r1 = __builtin_nontemporal_load(reinterpret_cast<const long long*>(p) + 1);
  } else {
    const double2 d = *reinterpret_cast<const double2*>(p);
    r0 = __double_as_longlong(d.x); r1 = __double_as_longlong(d.y);
  }
  v0 = make_double2((double)__int_as_float((int)(r0 & 0xFFFFFFFFll)), (double)__int_as_float((int)(r0 >> 32)));
  v1 = make_double2((double)__int_as_float((int)(r1 & 0xFFFFFFFFll)), (double)__int_as_float((int)(r1 >> 32)));
}

template <bool NT>
__device__ __forceinline__ void st(cplx* p, cplx v) {
  if (NT) {
    __builtin_nontemporal_store(v.x, &p->x);
    __builtin_nontemporal_store(v.y, &p->y);
  } else {
    *p = v;
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// Kernel A lane layout, either storage precision.  A lane owns CW consecutive column entries of one matrix row, i.e. ONE
// 16-byte fragment of each matrix and of each vector it needs:
//     fp64: CW = 1  -- lane (r, c) holds M[r][c] and x[c]                      (nc^2 lanes per site)
//     fp32: CW = 2  -- lane (r, h) holds M[r][2h..2h+1] and x[2h..2h+1]        (nc^2/2 lanes per site; nc = 1: CW = 1, 8 bytes)
// so every matrix load of a wavefront is one coalesced 1-KiB segment in BOTH precisions (with 8-byte loads the fp32
// kernel would issue the same number of load instructions for half the bytes).  For Wilson fp32 (nc = 2) a lane then
// holds a whole matrix row and the whole site vector: the row sum needs no cross-lane step at all.  The kernel computes
// in the storage type T (the fine operator is the one place fp32 ARITHMETIC is used: SURVEY 8c states 5e-6 per apply).
template <typename T, int NC>
struct KA {
  typedef typename CStore<T>::type ct;
  static constexpr int CW = (sizeof(T) == 4 && NC % 2 == 0) ? 2 : 1;
  static constexpr int LPR = NC / CW;          // lanes per matrix row
  static constexpr int E = NC * LPR;           // lanes per site
  struct Frag { ct v[CW]; };
};

template <typename T> __device__ __forceinline__ typename CStore<T>::type czero() { typename CStore<T>::type z; z.x = (T)0; z.y = (T)0; return z; }
template <typename CT> __device__ __forceinline__ void cmac_t(CT& acc, CT a, CT b) {
  acc.x = fma(a.x, b.x, acc.x);
  acc.x = fma(-a.y, b.y, acc.x);
  acc.y = fma(a.x, b.y, acc.y);
  acc.y = fma(a.y, b.x, acc.y);
}
__device__ __forceinline__ float lane_xor1(float v) { return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0xB1, 0xF, 0xF, true)); }
__device__ __forceinline__ float lane_xor2(float v) { return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0x4E, 0xF, 0xF, true)); }

typedef float v4f __attribute__((ext_vector_type(4)));

// fragment `idx` (in units of CW elements) of a T-typed array
template <typename T, int NC, bool NT>
__device__ __forceinline__ typename KA<T, NC>::Frag ld_frag(const void* base, long idx) {
  typename KA<T, NC>::Frag f;
  if constexpr (sizeof(T) == 8) {
    f.v[0] = ld<NT>(reinterpret_cast<const cplx*>(base) + idx);
  } else if constexpr (KA<T, NC>::CW == 2) {
    const v4f* p = reinterpret_cast<const v4f*>(base) + idx;
    const v4f r = NT ? __builtin_nontemporal_load(p) : *p;
    f.v[0].x = r.x; f.v[0].y = r.y; f.v[1].x = r.z; f.v[1].y = r.w;
  } else {
    const long long* p = reinterpret_cast<const long long*>(base) + idx;
    const long long raw = NT ? __builtin_nontemporal_load(p) : *p;
    f.v[0].x = __int_as_float((int)(raw & 0xFFFFFFFFll));
    f.v[0].y = __int_as_float((int)(raw >> 32));
  }
  return f;
}
template <typename T, int NC> __device__ __forceinline__ typename KA<T, NC>::Frag zero_frag() {
  typename KA<T, NC>::Frag f;
#pragma unroll
  for (int w = 0; w < KA<T, NC>::CW; w++) f.v[w] = czero<T>();
  return f;
}
template <typename T, int NC>
__device__ __forceinline__ void fmac(typename CStore<T>::type& acc, const typename KA<T, NC>::Frag& m, const typename KA<T, NC>::Frag& x) {
#pragma unroll
  for (int w = 0; w < KA<T, NC>::CW; w++) cmac_t(acc, m.v[w], x.v[w]);
}
// sum over the LPR lanes of a matrix row (adjacent lanes)
template <typename T, int NC> __device__ __forceinline__ void row_sum(typename CStore<T>::type& acc) {
  if (KA<T, NC>::LPR >= 2) { acc.x += lane_xor1(acc.x); acc.y += lane_xor1(acc.y); }
  if (KA<T, NC>::LPR >= 4) { acc.x += lane_xor2(acc.x); acc.y += lane_xor2(acc.y); }
}
// the shift coefficient on the diagonal, as a fragment: shift +- eo_shift +- dof_shift at column r (stencil_2d.h:890-908)
template <typename T, int NC>
__device__ __forceinline__ typename KA<T, NC>::Frag shift_frag(const StencilArgs& a, bool do_shift, int p, int r, int c0) {
  typename KA<T, NC>::Frag sh = zero_frag<T, NC>();
  if (do_shift) {
    const double sg = p ? -1.0 : 1.0;
    const double dg = (NC % 2 == 0) ? ((r < NC / 2) ? 1.0 : -1.0) : 0.0;
#pragma unroll
    for (int w = 0; w < KA<T, NC>::CW; w++)
      if (c0 + w == r) {
        sh.v[w].x = (T)(a.shift[0] + sg * a.eo_shift[0] + dg * a.dof_shift[0]);
        sh.v[w].y = (T)(a.shift[1] + sg * a.eo_shift[1] + dg * a.dof_shift[1]);
      }
  }
  return sh;
}
template <typename T, bool NTS>
__device__ __forceinline__ void st_elem(void* base, long i, typename CStore<T>::type v) {
  typedef typename CStore<T>::type ct;
  ct* p = reinterpret_cast<ct*>(base) + i;
  if (NTS) { __builtin_nontemporal_store(v.x, &p->x); __builtin_nontemporal_store(v.y, &p->y); }
  else *p = v;
}

template <typename T, int NC, bool NT, bool NTS>
__global__ __launch_bounds__(BLOCK) void k_stencil_elem(const StencilArgs a) {
  typedef KA<T, NC> K;
  typedef typename K::ct ct;
  typedef typename K::Frag Frag;
  constexpr int E = K::E, CW = K::CW, FPS = NC * NC / CW, VPS = NC / CW;   // fragments per site: matrix, vector
  const int e = threadIdx.x % E;
  const int r = e / K::LPR, c0 = (e % K::LPR) * CW;
  const int vf = e % K::LPR;            // this lane's vector fragment within a site
  const int j = blockIdx.x * (BLOCK / E) + threadIdx.x / E;
  if (j >= a.hr) return;   // whole site groups leave together (E divides BLOCK)

  for (int row = blockIdx.y; row < a.nrows; row += gridDim.y) {
    const int p = (a.par_count == 2) ? (row & 1) : a.par_first;
    const int y = (a.par_count == 2) ? (row >> 1) : row;
    const bool do_clover = a.clover && ((a.pieces >> p) & 1u);
    const unsigned hop_mask = a.hopping ? ((a.pieces >> (2 + 4 * p)) & 0xFu) : 0u;
    const bool do_shift = (a.pieces >> (10 + p)) & 1u;
    const bool do_zero = (a.pieces >> (12 + p)) & 1u;

    const long site = (long)p * a.half_vol + (long)y * a.hr + j;
    const long opp = (long)(1 - p) * a.half_vol;
    const int s = (y + p) & 1;
    int jp = j + s;     if (jp == a.hr) jp = 0;
    int jm = j + s - 1; if (jm < 0) jm = a.hr - 1;
    const int yp = (y + 1 == a.Ly) ? 0 : y + 1;
    const int ym = (y == 0) ? a.Ly - 1 : y - 1;
    long nb[4];
    nb[0] = opp + (long)y * a.hr + jp;
    nb[1] = opp + (long)yp * a.hr + j;
    nb[2] = opp + (long)y * a.hr + jm;
    nb[3] = opp + (long)ym * a.hr + j;

    // Stencil matrices: one coalesced 16-byte fragment per lane per matrix; kept in registers
    // across all right-hand sides.
    Frag m[5];
    m[4] = do_clover ? ld_frag<T, NC, NT>(a.clover, site * FPS + e) : zero_frag<T, NC>();
#pragma unroll
    for (int d = 0; d < 4; d++)
      m[d] = ((hop_mask >> d) & 1u) ? ld_frag<T, NC, NT>(a.hopping, ((long)d * a.size_cm) / CW + site * FPS + e) : zero_frag<T, NC>();
    const Frag sh = shift_frag<T, NC>(a, do_shift, p, r, c0);
    const bool need_own = do_clover || do_shift;

    for (int k = 0; k < a.nrhs; k++) {
      const ct* x = reinterpret_cast<const ct*>(a.rhs) + rhs_offset(a, k);
      ct* out = reinterpret_cast<ct*>(a.lhs) + rhs_offset(a, k);
      Frag xv[5];
#pragma unroll
      for (int d = 0; d < 4; d++)
        xv[d] = ((hop_mask >> d) & 1u) ? ld_frag<T, NC, false>(x, (QMG_ABLATE(a, 1) ? site : nb[d]) * VPS + vf) : zero_frag<T, NC>();
      xv[4] = need_own ? ld_frag<T, NC, false>(x, site * VPS + vf) : zero_frag<T, NC>();
      if (QMG_ABLATE(a, 4)) { for (int d = 0; d < 5; d++) for (int w = 0; w < CW; w++) { xv[d].v[w].x = (T)(1.0 + c0 + w); xv[d].v[w].y = (T)0.5; } }

      ct acc = czero<T>();
      fmac<T, NC>(acc, m[4], xv[4]);                       // clover first, as the reference does
#pragma unroll
      for (int d = 0; d < 4; d++) fmac<T, NC>(acc, m[d], xv[d]);
      fmac<T, NC>(acc, sh, xv[4]);
      row_sum<T, NC>(acc);

      if (QMG_ABLATE(a, 2) && acc.x != (T)1.2345e30) continue;
      if (c0 == 0) {
        if (!do_zero) { const ct o = out[site * NC + r]; acc.x += o.x; acc.y += o.y; }
        st_elem<T, NTS>(out, site * NC + r, acc);
      }
    }
  }
}

// ------------------------------------------------------------------------------------------
// Kernel A2 (nc = 1, 2, 4; both parities active): one lane group owns the EVEN and the ODD site
// of half-row column j on ROWS consecutive rows.  The two sites (y,j) of opposite parity are
// mutual x-neighbours (x = 2j and 2j+1, in an order set by y&1) and the rows share their
// y-neighbours, so the right-hand side is loaded into registers once for all 2*ROWS outputs:
// 2(ROWS+2) column values + 2 ROWS side values instead of 10 ROWS.  More importantly each
// wavefront now streams 2*ROWS*5 matrix fragments per lane (ROWS=2: 32 KiB of loads in flight per
// wave), which takes the launch out of the "a million 6-KiB waves" regime where wave dispatch,
// not HBM, sets the pace (tools/membw2.hip: 5.3 TB/s at 1M blocks vs 6.4-6.7 TB/s at 64K).
// ------------------------------------------------------------------------------------------
// NORM: the kernel also leaves |lhs_k|^2 of what it stored, as one partial per (block, system) in a.norm_part (summed in a
// fixed order by k_apply_norm_final) -- the residual norm of a Krylov step without re-reading the vector it has just written
// (16 of a staggered step's 56 B/site/rhs).  Inside the loop over the systems a lane only adds to its own LDS slot
// ([system][thread], dynamic shared memory): a wavefront reduction per system there (four ds_bpermute round trips in the
// dependent chain of every iteration) cost 0.15 ms on a 1.00 ms apply; the cross-lane sums happen once, after the loop.
// Lane groups past the end of the half row stay (on the last column, storing nothing) so that the block-wide steps see
// every thread.
// PF (batches): the right-hand side of system k+1 is requested before system k is multiplied, so a wavefront's loads stay in
// flight through its arithmetic and stores (the kernel sits at 2 waves/SIMD either way: 176 -> 2xx VGPRs).
template <typename T, int NC, int ROWS, bool NT, bool NTS, bool NORM = false, bool PF = false>
__global__ __launch_bounds__(BLOCK) void k_stencil_pair(const StencilArgs a) {
  typedef KA<T, NC> K;
  typedef typename K::ct ct;
  typedef typename K::Frag Frag;
  constexpr int E = K::E, CW = K::CW, FPS = NC * NC / CW, VPS = NC / CW;
  const int e = threadIdx.x % E;
  const int r = e / K::LPR, c0 = (e % K::LPR) * CW;
  const int vf = e % K::LPR;
  int j = blockIdx.x * (BLOCK / E) + threadIdx.x / E;
  bool live = true;
  if (j >= a.hr) {
    if (!NORM) return;
    live = false; j = a.hr - 1;
  }
  extern __shared__ double norm_sm[];
  if (NORM)
    for (int k = 0; k < a.nrhs; k++) norm_sm[k * BLOCK + threadIdx.x] = 0.0;
  const int ngroups = a.Ly / ROWS;

  bool do_clover[2], do_shift[2], do_zero[2];
  unsigned hop_mask[2];
  Frag sh[2];
#pragma unroll
  for (int p = 0; p < 2; p++) {
    do_clover[p] = a.clover && ((a.pieces >> p) & 1u);
    hop_mask[p] = a.hopping ? ((a.pieces >> (2 + 4 * p)) & 0xFu) : 0u;
    do_shift[p] = (a.pieces >> (10 + p)) & 1u;
    do_zero[p] = (a.pieces >> (12 + p)) & 1u;
    sh[p] = shift_frag<T, NC>(a, do_shift[p], p, r, c0);
  }
  const bool any_hop = (hop_mask[0] | hop_mask[1]) != 0u;
  int jl = j - 1; if (jl < 0) jl = a.hr - 1;
  int jr = j + 1; if (jr == a.hr) jr = 0;

  for (int grp = blockIdx.y; grp < ngroups; grp += gridDim.y) {
    const int y0 = grp * ROWS;

    // ---- stencil matrices: 2*ROWS*5 coalesced 16-byte fragments per lane
    Frag m[ROWS][2][5];
#pragma unroll
    for (int rr = 0; rr < ROWS; rr++)
#pragma unroll
      for (int p = 0; p < 2; p++) {
        const long site = (long)p * a.half_vol + (long)(y0 + rr) * a.hr + j;
        m[rr][p][4] = do_clover[p] ? ld_frag<T, NC, NT>(a.clover, site * FPS + e) : zero_frag<T, NC>();
#pragma unroll
        for (int d = 0; d < 4; d++)
          m[rr][p][d] = ((hop_mask[p] >> d) & 1u) ? ld_frag<T, NC, NT>(a.hopping, ((long)d * a.size_cm) / CW + site * FPS + e) : zero_frag<T, NC>();
      }

    // one system's right-hand side: rows y0-1 .. y0+ROWS at column j, both parities, and the one x-neighbour per row that
    // is not the partner site
    struct XF { Frag Ec[ROWS + 2], Oc[ROWS + 2], Es[ROWS], Os[ROWS]; };
    auto load_x = [&](XF& v, int k) {
      const ct* xe = reinterpret_cast<const ct*>(a.rhs) + rhs_offset_a(a, k);   // even half
      const ct* xo = xe + a.half_vol * NC;                                    // odd half
#pragma unroll
      for (int t = 0; t < ROWS + 2; t++) {
        int yy = y0 - 1 + t;
        if (yy < 0) yy = a.Ly - 1;
        if (yy >= a.Ly) yy -= a.Ly;
        const bool edge = (t == 0 || t == ROWS + 1);
        if (!edge || any_hop) {
          v.Ec[t] = ld_frag<T, NC, false>(xe, ((long)yy * a.hr + j) * VPS + vf);
          v.Oc[t] = ld_frag<T, NC, false>(xo, ((long)yy * a.hr + j) * VPS + vf);
        } else {
          v.Ec[t] = v.Oc[t] = zero_frag<T, NC>();
        }
      }
#pragma unroll
      for (int rr = 0; rr < ROWS; rr++) {
        const int y = y0 + rr;
        const int se = y & 1;                    // even site: x = 2j + se ; odd site: x = 2j + 1 - se
        if (any_hop) {
          v.Os[rr] = ld_frag<T, NC, false>(xo, ((long)y * a.hr + (se ? jr : jl)) * VPS + vf);
          v.Es[rr] = ld_frag<T, NC, false>(xe, ((long)y * a.hr + (se ? jl : jr)) * VPS + vf);
        } else {
          v.Os[rr] = v.Es[rr] = zero_frag<T, NC>();
        }
      }
    };

    XF cur;
    if (PF) load_x(cur, 0);
    for (int k = 0; k < a.nrhs; k++) {
      ct* out = reinterpret_cast<ct*>(a.lhs) + rhs_offset_a(a, k);
      double nrm = 0.0;
      XF nxt;
      if (!PF) load_x(cur, k);
      else {
        if (k + 1 < a.nrhs) load_x(nxt, k + 1);
        __builtin_amdgcn_sched_barrier(0);       // keep the scheduler from sinking the prefetch below the arithmetic
      }
      const Frag (&Ec)[ROWS + 2] = cur.Ec, (&Oc)[ROWS + 2] = cur.Oc;
      const Frag (&Es)[ROWS] = cur.Es, (&Os)[ROWS] = cur.Os;

#pragma unroll
      for (int rr = 0; rr < ROWS; rr++) {
        const int y = y0 + rr;
        const int se = y & 1;
#pragma unroll
        for (int p = 0; p < 2; p++) {
          // neighbours of the parity-p site (y, j); s = (y + p) & 1
          const int s = p ? (1 - se) : se;
          const Frag own = p ? Oc[rr + 1] : Ec[rr + 1];
          const Frag partner = p ? Ec[rr + 1] : Oc[rr + 1];     // opposite parity, same (y, j)
          const Frag side = p ? Es[rr] : Os[rr];                // opposite parity, (y, j + (s ? +1 : -1))
          Frag xv[4];
          xv[0] = s ? side : partner;                           // +x: (y, j + s)
          xv[2] = s ? partner : side;                           // -x: (y, j + s - 1)
          xv[1] = p ? Ec[rr + 2] : Oc[rr + 2];                  // +y
          xv[3] = p ? Ec[rr] : Oc[rr];                          // -y
          // values this parity does not ask for may be uninitialised memory: never let them into the sum
          const Frag zero = zero_frag<T, NC>();
          const Frag own_u = (do_clover[p] || do_shift[p]) ? own : zero;
          ct acc = czero<T>();
          fmac<T, NC>(acc, m[rr][p][4], own_u);
#pragma unroll
          for (int d = 0; d < 4; d++) fmac<T, NC>(acc, m[rr][p][d], ((hop_mask[p] >> d) & 1u) ? xv[d] : zero);
          fmac<T, NC>(acc, sh[p], own_u);
          row_sum<T, NC>(acc);
          const bool touch = do_clover[p] || hop_mask[p] || do_shift[p] || do_zero[p];
          if (c0 == 0 && touch && live) {
            const long o = ((long)p * a.half_vol + (long)y * a.hr + j) * NC + r;
            if (!do_zero[p]) { const ct prev = out[o]; acc.x += prev.x; acc.y += prev.y; }
            st_elem<T, NTS>(out, o, acc);
            if (NORM) { nrm = fma((double)acc.x, (double)acc.x, nrm); nrm = fma((double)acc.y, (double)acc.y, nrm); }
          }
        }
      }
      if (NORM) norm_sm[k * BLOCK + threadIdx.x] += nrm;
      if (PF && k + 1 < a.nrhs) cur = nxt;
    }
  }
  if (NORM) {
    __syncthreads();
    const int lane = threadIdx.x & (WAVE - 1);
    for (int k = threadIdx.x / WAVE; k < a.nrhs; k += BLOCK / WAVE) {
      double t = norm_sm[k * BLOCK + lane];
#pragma unroll
      for (int w = 1; w < BLOCK / WAVE; w++) t += norm_sm[k * BLOCK + w * WAVE + lane];
      t = wave_sum(t);
      if (lane == 0) a.norm_part[(long)k * ((long)gridDim.y * gridDim.x) + (long)blockIdx.y * gridDim.x + blockIdx.x] = t;   // [system][block]
    }
  }
}

// the partials of system q (contiguous: [system][block]), summed in a fixed order
__global__ __launch_bounds__(BLOCK) void k_apply_norm_final(const double* __restrict__ part, long nparts, double* __restrict__ out) {
  __shared__ double sm[BLOCK / WAVE];
  const int q = blockIdx.x;
  double t = 0.0;
  for (long i = threadIdx.x; i < nparts; i += BLOCK) t += part[(long)q * nparts + i];
  t = wave_sum(t);
  if ((threadIdx.x & (WAVE - 1)) == 0) sm[threadIdx.x / WAVE] = t;
  __syncthreads();
  if (threadIdx.x == 0) {
    double r = sm[0];
#pragma unroll
    for (int w = 1; w < BLOCK / WAVE; w++) r += sm[w];
    out[q] = r;
  }
}

// ------------------------------------------------------------------------------------------
// Kernel B: any nc (coarse operators, nc = 8, 24, ...).  One block owns S consecutive sites of
// one row.  Per piece (clover, 4 directions) the block copies the S matrices (S nc^2 x 16 B,
// contiguous in the reference layout) global -> registers -> LDS with fully coalesced 16-byte
// loads, software-pipelined one piece ahead, and the S neighbour vectors likewise.  Thread
// (s, r, h) then accumulates the h-th slice of sum_c M[s][r][c] x[s][c] out of LDS (rows padded
// by one element when nc is even so that 16 lanes of a ds_read_b128 hit 64 distinct banks), the
// H slices are summed through LDS, and one thread per (s, r) writes the result.
// The operation is HBM-bound (AI ~ 0.5 flop/B for one right-hand side): all that
// matters is that the matrix stream is coalesced and deep enough in flight.
// ------------------------------------------------------------------------------------------
struct GenLayout {
  int S;        // sites per block
  int H;        // c-slices per row
  int rs;       // padded LDS row stride (complex elements)
  int mat_elems;   // S * nc * nc
  int per_thread;  // ceil(mat_elems / BLOCK)
};

constexpr int GEN_MAX_PER_THREAD = 12;   // register-staged matrix elements per thread per piece

// KR = right-hand sides per pass: the matrix tile parked in LDS is used for KR vectors (KR accumulators per thread), so a
// batch reads the matrices once per KR systems for ANY nc -- the vector-FMA counterpart of kernel C, and the better one
// where the 16x16 MFMA tile would be mostly padding (nc = 8: 1024^2, 8 rhs 2.0 ms on the matrix cores).
// EPI (KR = 1 only): the apply epilogue of qmg_common.h, a COMPILE-TIME switch -- as a run-time branch it cost every launch ~9 VGPRs and,
// for several tile shapes, a wavefront of occupancy.
template <int PT, bool M32, int KR, bool V32, bool EPI = false>
__global__ __launch_bounds__(BLOCK) void k_stencil_gen(const StencilArgs a, const int nc, const GenLayout L) {
  extern __shared__ __align__(16) unsigned char smem_raw[];
  cplx* mlds = reinterpret_cast<cplx*>(smem_raw);                    // [S*nc rows][rs]
  cplx* xlds = mlds + (size_t)L.S * nc * L.rs;                        // [KR][S][nc]
  cplx* red = xlds + (size_t)KR * L.S * nc;                           // [H][S*nc]

  const int tid = threadIdx.x;
  const int rows = L.S * nc;             // (s, r) pairs in this block
  const int h = tid / rows;              // slice id (threads beyond H*rows idle in the compute phase)
  const int sr = tid - h * rows;
  const bool worker = h < L.H;
  const int s_of = sr / nc;
  const int r_of = sr - s_of * nc;
  const int cchunk = (nc + L.H - 1) / L.H;
  const int c0 = h * cchunk;
  const int c1 = (c0 + cchunk < nc) ? c0 + cchunk : nc;

  const int j0 = blockIdx.x * L.S;
  const int nsite = (a.hr - j0 < L.S) ? a.hr - j0 : L.S;    // ragged last tile
  const long nc2 = (long)nc * nc;
  double edots[3] = {0.0, 0.0, 0.0};   // MR dots of the epilogue (EPI instantiations: one system per launch)

  for (int row = blockIdx.y; row < a.nrows; row += gridDim.y) {
    const int p = (a.par_count == 2) ? (row & 1) : a.par_first;
    const int y = (a.par_count == 2) ? (row >> 1) : row;
    const bool do_clover = a.clover && ((a.pieces >> p) & 1u);
    const unsigned hop_mask = a.hopping ? ((a.pieces >> (2 + 4 * p)) & 0xFu) : 0u;
    const bool do_shift = (a.pieces >> (10 + p)) & 1u;
    const bool do_zero = (a.pieces >> (12 + p)) & 1u;
    const unsigned piece_mask = hop_mask | (do_clover ? 16u : 0u);   // bit 4 = clover

    const long site0 = (long)p * a.half_vol + (long)y * a.hr + j0;
    const long opp = (long)(1 - p) * a.half_vol;
    const int s = (y + p) & 1;
    const int yp = (y + 1 == a.Ly) ? 0 : y + 1;
    const int ym = (y == 0) ? a.Ly - 1 : y - 1;
    cplx e_ov = cmake(0.0, 0.0), e_dv = cmake(0.0, 0.0);   // the epilogue's operands of this thread's output element, requested up front
    if (EPI && h == 0 && s_of < nsite) {
      const long o = rhs_offset(a, 0) + (site0 + s_of) * nc + r_of;
      if (a.epi.other) e_ov = ldv<V32>(a.epi.other, o);
      if (a.epi.dotv) e_dv = (a.epi.dotv == a.epi.other) ? e_ov : ldv<V32>(a.epi.dotv, o);
    }

    for (int k0 = 0; k0 < a.nrhs; k0 += KR) {
      const int nk = (a.nrhs - k0 < KR) ? a.nrhs - k0 : KR;
      cplx acc[KR];
#pragma unroll
      for (int kk = 0; kk < KR; kk++) acc[kk] = cmake(0.0, 0.0);

      // piece order: clover (4), +x, +y, -x, -y  -- the reference's accumulation order
      const int order[5] = {4, 0, 1, 2, 3};
      // fp32-stored matrices with even nc: a lane loads PAIRS of elements (16 B per load, as in the fp64 stream) -- with 8-B
      // loads the same number of load instructions moved half the bytes and the apply got no faster
      constexpr int PTS = PT + (PT & 1);
      const bool pairs = M32 && !(nc & 1);
      typename MRaw<M32>::type stage[PTS];            // storage form (widened when parked)
      typename XRaw<V32>::type xstage[KR];
#pragma unroll
      for (int kk = 0; kk < KR; kk++) xstage[kk] = zero_raw<V32>();
      int cur = -1;
      // find first active piece and prefetch it
      int oi = 0;
      while (oi < 5 && !((piece_mask >> order[oi]) & 1u)) oi++;
      auto prefetch = [&](int piece) {
        const cplx* mbase = (piece == 4) ? a.clover : a.hopping;                 // (element offsets, so that the same
        long moff = (piece == 4) ? site0 * nc2 : (long)piece * a.size_cm + site0 * nc2;   //  code serves both matrix widths)
        if (QMG_ABLATE(a, 16) && (piece == 2 || piece == 3)) {
          // diagnostic (wrong arithmetic, right access pattern): what the backward hops would cost if they re-read the
          // neighbour's FORWARD link (gamma5-hermitian link compression) instead of streaming their own array
          long nsite0 = (piece == 2) ? opp + (long)y * a.hr + (j0 + s - 1 < 0 ? 0 : j0 + s - 1) : opp + (long)ym * a.hr + j0;
          if (nsite0 + nsite > 2 * a.half_vol) nsite0 = 2 * a.half_vol - nsite;
          moff = (long)(piece - 2) * a.size_cm + nsite0 * nc2;
        }
        const int lim = nsite * (int)nc2;
        if (pairs) {
#pragma unroll
          for (int q = 0; q < PTS / 2; q++) {
            const int el = 2 * (tid + q * BLOCK);
            stage[2 * q] = zero_mraw<M32>(); stage[2 * q + 1] = zero_mraw<M32>();
            if (el < lim) {
              if constexpr (M32) {   // 16 bytes: two raw elements
                const long long* pp = reinterpret_cast<const long long*>(mbase) + moff + el;
                stage[2 * q] = __builtin_nontemporal_load(pp);
                stage[2 * q + 1] = __builtin_nontemporal_load(pp + 1);
              }
            }
          }
        } else {
#pragma unroll
          for (int q = 0; q < PT; q++) {
            const int el = tid + q * BLOCK;
            stage[q] = zero_mraw<M32>();
            if (el < lim) stage[q] = ldm_raw<M32, true>(mbase, moff + el);
          }
        }
        // neighbour vector element for (site, c) = tid / nc, tid % nc
        if (tid < nsite * nc) {
          const int sl = tid / nc, cc = tid - sl * nc;
          const int j = j0 + sl;
          long nbsite;
          if (piece == 4) nbsite = site0 + sl;
          else if (piece == 0) { int jp = j + s; if (jp == a.hr) jp = 0; nbsite = opp + (long)y * a.hr + jp; }
          else if (piece == 1) nbsite = opp + (long)yp * a.hr + j;
          else if (piece == 2) { int jm = j + s - 1; if (jm < 0) jm = a.hr - 1; nbsite = opp + (long)y * a.hr + jm; }
          else nbsite = opp + (long)ym * a.hr + j;
          // a slab's rows -1 / Ly: the opposite-parity row of the halo buffer (row-uniform choice)
          const bool hi = piece == 1 && a.halo_hi && y + 1 == a.Ly, lo = piece == 3 && a.halo_lo && y == 0;
          const long hsite = (long)(1 - p) * a.hr + j;
#pragma unroll
          for (int kk = 0; kk < KR; kk++)
            if (kk < nk) {
              const int ks = system_index(a, k0 + kk);
              if (hi) xstage[kk] = ldv_raw<V32>(a.halo_hi, (long)ks * a.halo_stride + hsite * nc + cc);
              else if (lo) xstage[kk] = ldv_raw<V32>(a.halo_lo, (long)ks * a.halo_stride + hsite * nc + cc);
              else xstage[kk] = ldv_raw<V32>(a.rhs, rhs_offset(a, k0 + kk) + nbsite * nc + cc);
            }
        }
      };
      if (oi < 5) { cur = order[oi]; prefetch(cur); }

      while (cur >= 0) {
        __syncthreads();   // previous compute finished reading LDS
        // registers -> LDS (padded rows)
        if (pairs) {
#pragma unroll
          for (int q = 0; q < PTS / 2; q++) {
            const int el = 2 * (tid + q * BLOCK);
            if (el < L.mat_elems) {   // (mat_elems and nc even: the pair never straddles a row)
              const int rowi = el / nc, cc = el - rowi * nc;
              mlds[(size_t)rowi * L.rs + cc] = widen_mraw<M32>(stage[2 * q]);
              mlds[(size_t)rowi * L.rs + cc + 1] = widen_mraw<M32>(stage[2 * q + 1]);
            }
          }
        } else {
#pragma unroll
          for (int q = 0; q < PT; q++) {
            const int el = tid + q * BLOCK;
            if (el < L.mat_elems) {
              const int rowi = el / nc, cc = el - rowi * nc;
              mlds[(size_t)rowi * L.rs + cc] = widen_mraw<M32>(stage[q]);
            }
          }
        }
        if (tid < L.S * nc) {
#pragma unroll
          for (int kk = 0; kk < KR; kk++) xlds[kk * rows + tid] = widen_raw<V32>(xstage[kk]);
        }
        // issue the next piece's global loads before computing on this one
        int nxt = -1;
        oi++;
        while (oi < 5 && !((piece_mask >> order[oi]) & 1u)) oi++;
        if (oi < 5) { nxt = order[oi]; prefetch(nxt); }
        __syncthreads();
        if (worker && s_of < nsite && !QMG_ABLATE(a, 32)) {
          const cplx* mrow = mlds + (size_t)sr * L.rs;
          const cplx* xs = xlds + s_of * nc;
          for (int cc = c0; cc < c1; cc++) {
            const cplx m = mrow[cc];          // one LDS read of the matrix element serves all KR right-hand sides
#pragma unroll
            for (int kk = 0; kk < KR; kk++) cmac(acc[kk], m, xs[kk * rows + cc]);
          }
        }
        if (QMG_ABLATE(a, 32)) acc[0] = cadd(acc[0], widen_mraw<M32>(stage[0]));   // diagnostic: no LDS reads / FMAs, loads kept alive
        cur = nxt;
      }

      // shift term needs the own-site vector
      if (do_shift && worker && h == 0 && s_of < nsite) {
        const double sg = p ? -1.0 : 1.0;
        const double dg = (nc % 2 == 0) ? ((r_of < nc / 2) ? 1.0 : -1.0) : 0.0;
        const cplx sh = cmake(a.shift[0] + sg * a.eo_shift[0] + dg * a.dof_shift[0],
                              a.shift[1] + sg * a.eo_shift[1] + dg * a.dof_shift[1]);
#pragma unroll
        for (int kk = 0; kk < KR; kk++)
          if (kk < nk) cmac(acc[kk], sh, ldv<V32>(a.rhs, rhs_offset(a, k0 + kk) + (site0 + s_of) * nc + r_of));
      }
      // sum the H slices, one right-hand side at a time through the same LDS buffer
#pragma unroll
      for (int kk = 0; kk < KR; kk++) {
        if (kk >= nk) break;
        __syncthreads();
        if (worker) red[(size_t)h * rows + sr] = acc[kk];
        __syncthreads();
        if (h == 0 && s_of < nsite) {
          cplx t = red[sr];
          for (int hh = 1; hh < L.H; hh++) t = cadd(t, red[(size_t)hh * rows + sr]);
          const long o = rhs_offset(a, k0 + kk) + (site0 + s_of) * nc + r_of;
          if (!do_zero) t = cadd(ldv<V32>(a.lhs, o), t);
          if (EPI) t = epilogue_value<V32>(a.epi, e_ov, e_dv, t, edots);
          stv<V32>(a.lhs, o, t);
        }
      }
    }
  }
  if (EPI && a.epi.dotv) epilogue_store_partials(a.epi, edots);
}

// Kernel B32 (opt-in complex<float> matrix storage, even nc): kernel B with the tile kept in fp32 end to end -- 16-B
// loads carry two matrix elements, the staging registers and the LDS tile hold raw float pairs (half the registers, half
// the LDS: twice the resident blocks), and an element is widened to fp64 only when it is multiplied.  PP = staged PAIRS per
// thread.  Row stride nc + 2 floats-pairs: even (16-B aligned pair stores) and conflict-free for the 8-byte row reads.
// M16: the matrices are stored as complex<half> (qmg_stencil_apply_mat16; nc a multiple of 4): a 16-B load carries FOUR elements (PP = staged quads
// per thread), which are widened to complex<float> when they are parked -- the LDS tile and everything behind it are those of the fp32 form.
template <int PP, int KR, bool V32, bool EPI = false, bool M16 = false>
__global__ __launch_bounds__(BLOCK) void k_stencil_gen32(const StencilArgs a, const int nc, const GenLayout L) {
  extern __shared__ __align__(16) unsigned char smem_raw[];
  const int rs32 = nc + 2;
  float2* mlds = reinterpret_cast<float2*>(smem_raw);                                     // [S*nc rows][rs32] complex<float>
  cplx* xlds = reinterpret_cast<cplx*>(smem_raw + (((size_t)L.S * nc * rs32 * 8 + 15) & ~(size_t)15));   // [KR][S][nc]
  cplx* red = xlds + (size_t)KR * L.S * nc;                                               // [H][S*nc]

  const int tid = threadIdx.x;
  const int rows = L.S * nc;             // (s, r) pairs in this block
  const int h = tid / rows;              // slice id (threads beyond H*rows idle in the compute phase)
  const int sr = tid - h * rows;
  const bool worker = h < L.H;
  const int s_of = sr / nc;
  const int r_of = sr - s_of * nc;
  const int cchunk = (nc + L.H - 1) / L.H;
  const int c0 = h * cchunk;
  const int c1 = (c0 + cchunk < nc) ? c0 + cchunk : nc;

  const int j0 = blockIdx.x * L.S;
  const int nsite = (a.hr - j0 < L.S) ? a.hr - j0 : L.S;    // ragged last tile
  const long nc2 = (long)nc * nc;
  double edots[3] = {0.0, 0.0, 0.0};   // MR dots of the epilogue (EPI instantiations: one system per launch)

  for (int row = blockIdx.y; row < a.nrows; row += gridDim.y) {
    const int p = (a.par_count == 2) ? (row & 1) : a.par_first;
    const int y = (a.par_count == 2) ? (row >> 1) : row;
    const bool do_clover = a.clover && ((a.pieces >> p) & 1u);
    const unsigned hop_mask = a.hopping ? ((a.pieces >> (2 + 4 * p)) & 0xFu) : 0u;
    const bool do_shift = (a.pieces >> (10 + p)) & 1u;
    const bool do_zero = (a.pieces >> (12 + p)) & 1u;
    const unsigned piece_mask = hop_mask | (do_clover ? 16u : 0u);   // bit 4 = clover

    const long site0 = (long)p * a.half_vol + (long)y * a.hr + j0;
    const long opp = (long)(1 - p) * a.half_vol;
    const int s = (y + p) & 1;
    const int yp = (y + 1 == a.Ly) ? 0 : y + 1;
    const int ym = (y == 0) ? a.Ly - 1 : y - 1;
    cplx e_ov = cmake(0.0, 0.0), e_dv = cmake(0.0, 0.0);   // the epilogue's operands of this thread's output element, requested up front
    if (EPI && h == 0 && s_of < nsite) {
      const long o = rhs_offset(a, 0) + (site0 + s_of) * nc + r_of;
      if (a.epi.other) e_ov = ldv<V32>(a.epi.other, o);
      if (a.epi.dotv) e_dv = (a.epi.dotv == a.epi.other) ? e_ov : ldv<V32>(a.epi.dotv, o);
    }

    for (int k0 = 0; k0 < a.nrhs; k0 += KR) {
      const int nk = (a.nrhs - k0 < KR) ? a.nrhs - k0 : KR;
      cplx acc[KR];
#pragma unroll
      for (int kk = 0; kk < KR; kk++) acc[kk] = cmake(0.0, 0.0);

      // piece order: clover (4), +x, +y, -x, -y  -- the reference's accumulation order.  The active pieces (uniform over the block) are
      // walked by a loop the compiler unrolls, so the staging-register sets have compile-time indices: PF pieces are requested ahead of the
      // one that computes.  PF = 2 for the 16-bit storage: a piece is half the bytes of the fp32 form, so with one piece ahead a block
      // had half the bytes in flight and the kernel stopped at 0.61 of the HBM rate; two sets of quads cost what one set of pairs does.
      constexpr int PF = M16 ? 2 : 1;
      double2 stage[PF][PP];   // raw bits of two complex<float> (four complex<half>) each
      typename XRaw<V32>::type xstage[PF][KR];   // (storage form: widened when they are parked)
#pragma unroll
      for (int f = 0; f < PF; f++)
#pragma unroll
        for (int kk = 0; kk < KR; kk++) xstage[f][kk] = zero_raw<V32>();
      // bit oi of om: the oi-th piece of the order {clover, +x, +y, -x, -y} is active
      unsigned om = ((piece_mask >> 4) & 1u) | ((piece_mask & 0xFu) << 1);
      const int npc = __popc(om);
      int lst[5];
#pragma unroll
      for (int i = 0; i < 5; i++) { const int oi = om ? __ffs(om) - 1 : 0; lst[i] = (oi == 0) ? 4 : oi - 1; om &= om - 1; }
      auto prefetch = [&](int piece, int f) {
        const cplx* mbase = (piece == 4) ? a.clover : a.hopping;                 // (element offsets, so that the same
        long moff = (piece == 4) ? site0 * nc2 : (long)piece * a.size_cm + site0 * nc2;   //  code serves both matrix widths)
        const int lim = nsite * (int)nc2;
        const float2* m32 = reinterpret_cast<const float2*>(mbase) + moff;
        const unsigned* m16 = reinterpret_cast<const unsigned*>(mbase) + moff;   // complex<half>: 4 B per element
#pragma unroll
        for (int q = 0; q < PP; q++) {
          const int el = (M16 ? 4 : 2) * (tid + q * BLOCK);
          if (el < lim) {
            const double* pp = M16 ? reinterpret_cast<const double*>(m16 + el) : reinterpret_cast<const double*>(m32 + el);
            stage[f][q].x = __builtin_nontemporal_load(pp);
            stage[f][q].y = __builtin_nontemporal_load(pp + 1);
          } else stage[f][q] = make_double2(0.0, 0.0);
        }
        // neighbour vector element for (site, c) = tid / nc, tid % nc
        if (tid < nsite * nc) {
          const int sl = tid / nc, cc = tid - sl * nc;
          const int j = j0 + sl;
          long nbsite;
          if (piece == 4) nbsite = site0 + sl;
          else if (piece == 0) { int jp = j + s; if (jp == a.hr) jp = 0; nbsite = opp + (long)y * a.hr + jp; }
          else if (piece == 1) nbsite = opp + (long)yp * a.hr + j;
          else if (piece == 2) { int jm = j + s - 1; if (jm < 0) jm = a.hr - 1; nbsite = opp + (long)y * a.hr + jm; }
          else nbsite = opp + (long)ym * a.hr + j;
          // a slab's rows -1 / Ly: the opposite-parity row of the halo buffer (a row-uniform choice of base, stride and site: ONE load either way)
          const bool halo = (piece == 1 && a.halo_hi && y + 1 == a.Ly) || (piece == 3 && a.halo_lo && y == 0);
          const void* vbase = halo ? (piece == 1 ? a.halo_hi : a.halo_lo) : a.rhs;
          const long vstride = halo ? a.halo_stride : a.vec_stride;
          const long vsite = halo ? (long)(1 - p) * a.hr + j : nbsite;
#pragma unroll
          for (int kk = 0; kk < KR; kk++)
            if (kk < nk) xstage[f][kk] = ldv_raw<V32>(vbase, (long)system_index(a, k0 + kk) * vstride + vsite * nc + cc);
        }
      };
      // one piece: park set f (registers -> LDS), request piece `nextp` into the set just freed, compute
      auto do_piece = [&](int nextp, auto fc) {
        constexpr int f = decltype(fc)::value;
        __syncthreads();   // previous compute finished reading LDS
        // registers -> LDS (padded rows)
#pragma unroll
        for (int q = 0; q < PP; q++) {
          const int el = (M16 ? 4 : 2) * (tid + q * BLOCK);
          if (el < L.mat_elems) {   // (nc even: the pair never straddles a row; rs32 and cc even: 16-B aligned.  M16: nc % 4 == 0, the quad stays in its row)
            const int rowi = el / nc, cc = el - rowi * nc;
            if constexpr (M16) {
              typedef _Float16 h8 __attribute__((ext_vector_type(8)));
              typedef float f4 __attribute__((ext_vector_type(4)));
              const h8 hv = __builtin_bit_cast(h8, stage[f][q]);   // (re, im) x 4
              const f4 w0 = {(float)hv[0], (float)hv[1], (float)hv[2], (float)hv[3]}, w1 = {(float)hv[4], (float)hv[5], (float)hv[6], (float)hv[7]};
              *reinterpret_cast<f4*>(mlds + (size_t)rowi * rs32 + cc) = w0;
              *reinterpret_cast<f4*>(mlds + (size_t)rowi * rs32 + cc + 2) = w1;
            } else
              *reinterpret_cast<double2*>(mlds + (size_t)rowi * rs32 + cc) = stage[f][q];
          }
        }
        if (tid < L.S * nc) {
#pragma unroll
          for (int kk = 0; kk < KR; kk++) xlds[kk * rows + tid] = widen_raw<V32>(xstage[f][kk]);
        }
        // issue the global loads of the piece PF ahead (into the set just parked) before computing on this one
        if (nextp >= 0) prefetch(nextp, f);
        __syncthreads();
        if (worker && s_of < nsite && !QMG_ABLATE(a, 32)) {
          const float2* mrow = mlds + (size_t)sr * rs32;
          const cplx* xs = xlds + s_of * nc;
          for (int cc = c0; cc < c1; cc++) {
            const float2 mf = mrow[cc];       // one 8-B LDS read serves all KR right-hand sides; widened here
            const cplx m = make_double2((double)mf.x, (double)mf.y);
#pragma unroll
            for (int kk = 0; kk < KR; kk++) cmac(acc[kk], m, xs[kk * rows + cc]);
          }
        }
        if (QMG_ABLATE(a, 32)) acc[0] = cadd(acc[0], stage[f][0]);   // diagnostic: no LDS reads / FMAs, loads kept alive (raw bits)
      };
      if constexpr (PF == 1 && PP > 4) {
        // one piece ahead, large tiles (nc = 24: six staged pairs per thread): a plain loop.  Unrolled over the five pieces -- which is what the
        // smaller tiles get below: nc = 8, complex<float> vectors 185 -> 168 us per level-1 Schur hop of the C5 solve -- the compiler keeps every
        // piece's load addresses live: 89 -> 150 VGPRs, 171 with the epilogue, two wavefronts per SIMD instead of four, and the level-1 applies
        // of the C3 solve went 1.04 -> 1.15 ms.
        unsigned rest = ((piece_mask >> 4) & 1u) | ((piece_mask & 0xFu) << 1);
        auto pop = [&]() -> int { if (!rest) return -1; const int oi = __ffs(rest) - 1; rest &= rest - 1; return (oi == 0) ? 4 : oi - 1; };
        int cur = pop();
        if (cur >= 0) prefetch(cur, 0);
        while (cur >= 0) {
          const int nxt = pop();
          do_piece(nxt, std::integral_constant<int, 0>());
          cur = nxt;
        }
      } else {
#pragma unroll
        for (int i = 0; i < PF; i++)
          if (i < npc) prefetch(lst[i], i);
#pragma unroll
        for (int i = 0; i < 5; i++)
          if (i < npc) {
            const int nextp = (i + PF < npc) ? lst[i + PF] : -1;
            if (i % PF == 0) do_piece(nextp, std::integral_constant<int, 0>());
            else do_piece(nextp, std::integral_constant<int, PF - 1>());
          }
      }

      // shift term needs the own-site vector
      if (do_shift && worker && h == 0 && s_of < nsite) {
        const double sg = p ? -1.0 : 1.0;
        const double dg = (nc % 2 == 0) ? ((r_of < nc / 2) ? 1.0 : -1.0) : 0.0;
        const cplx sh = cmake(a.shift[0] + sg * a.eo_shift[0] + dg * a.dof_shift[0],
                              a.shift[1] + sg * a.eo_shift[1] + dg * a.dof_shift[1]);
#pragma unroll
        for (int kk = 0; kk < KR; kk++)
          if (kk < nk) cmac(acc[kk], sh, ldv<V32>(a.rhs, rhs_offset(a, k0 + kk) + (site0 + s_of) * nc + r_of));
      }
      // sum the H slices, one right-hand side at a time through the same LDS buffer
#pragma unroll
      for (int kk = 0; kk < KR; kk++) {
        if (kk >= nk) break;
        __syncthreads();
        if (worker) red[(size_t)h * rows + sr] = acc[kk];
        __syncthreads();
        if (h == 0 && s_of < nsite) {
          cplx t = red[sr];
          for (int hh = 1; hh < L.H; hh++) t = cadd(t, red[(size_t)hh * rows + sr]);
          const long o = rhs_offset(a, k0 + kk) + (site0 + s_of) * nc + r_of;
          if (!do_zero) t = cadd(ldv<V32>(a.lhs, o), t);
          if (EPI) t = epilogue_value<V32>(a.epi, e_ov, e_dv, t, edots);
          stv<V32>(a.lhs, o, t);
        }
      }
    }
  }
  if (EPI && a.epi.dotv) epilogue_store_partials(a.epi, edots);
}

// ---------------------------------------------------------------------------------------------------------------------
// Kernel C (nc in {8,12,16,24,32}, 2..16 right-hand sides per pass): the coarse apply as a real contraction on the f64
// matrix cores.  With k right-hand sides against one matrix read the per-site work is the (nc x nc) . (nc x k) product
//     out[r][k] (+)= sum_piece sum_c M_piece(x)[r][c] * X_k(nb_piece(x))[c]
// and the arithmetic intensity rises from 0.5 flop/B to ~0.5 k flop/B; 16 right-hand sides move 5 nc^2 + 32 nc complex
// per site instead of 16 (5 nc^2 + 2 nc).  One wavefront owns one output site.  v_mfma_f64_16x16x4_f64 tiles:
//     A (16 x 4)  = M[16 t + (lane&15)][4 s + (lane>>4)]       each lane's 16 B carries (re, im)
//     B (4 x 16)  = X_{lane&15}[4 s + (lane>>4)]                one right-hand side per MFMA column
//     C (16 x 16) : row = 16 t + 4 i + (lane>>4), column = lane&15 for accumulator register i      (f64 C/D map)
// A complex MAC is four real MFMAs (re += ar.br - ai.bi ; im += ar.bi + ai.br).  Rows / k-steps beyond nc and columns
// beyond the rhs count are fed zeros.
// Matrix stream: a site's piece is nc^2 contiguous complex; the wavefront reads it with fully coalesced non-temporal
// 1-KiB loads (lane-linear), parks it in its own LDS slice with odd row stride nc+1 (conflict-free operand reads), and
// pulls A fragments from there.  Operand-layout loads straight from HBM touch half a cache line per 4 lanes and ran at
// 4.5 TB/s with the MFMAs removed; the staged stream is what the 5.8 TB/s single-rhs kernels use.  The slice is private
// to the wavefront, so the write->read hand-off is a wavefront fence, not a block barrier; the global loads of piece
// p+2 are in flight while piece p+1 computes.  The own-site vector in B layout IS the shift term's operand in C layout
// (k-step s = 4 t + i holds row 16 t + 4 i + (lane>>4)); it is re-read from L2 in the epilogue.
typedef double v4d __attribute__((ext_vector_type(4)));
typedef float v4f32 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void wave_lds_handoff() {
  __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
  __builtin_amdgcn_wave_barrier();
}

// MODE 0: four real MFMAs per complex tile product (plain).
// MODE 1: at most 8 right-hand sides: columns 0-7 carry Re X_k, columns 8-15 Im X_k, so P = Re(M).[Xr|Xi] and
//         Q = Im(M).[Xr|Xi] are TWO MFMAs per tile product; the epilogue recombines re_k = P[k] - Q[k+8],
//         im_k = P[k+8] + Q[k] with one lane exchange (lane ^ 8).
// (A three-multiplication complex product for 9-16 right-hand sides was measured SLOWER than MODE 0 -- 4.09 vs 3.63 ms
// at 512^2, nc = 24, 16 rhs: the extra f64 adds and the third accumulator cost more than the saved MFMA -- and dropped.)
// The f64 matrix pipe sustains 48 TFLOP/s on this part (tools/mfma_f64_rate.hip), which at nc = 24 is 2.7 ms of plain
// MFMA work per 512^2 apply against 2.4 ms of HBM time -- the MFMA count, not the byte count, is what MODE 1 cuts.
// MODE 2 (9-16 right-hand sides, fp64, VL): the REAL form of the product -- [yr; yi] = [[Mr, -Mi], [Mi, Mr]] [xr; xi], a (2 nc x 2 nc) real
//         matrix against a (2 nc x 16) real right-hand side: ONE MFMA per 16 x 4 tile of it, each lane pulling the double it needs
//         (re or im of M[r][c], sign by quadrant) straight out of the complex LDS tile.  The MFMA count is 2 nc/16 (rounded up) x nc/2
//         per piece instead of MODE 0's 4 x ceil(nc/16) x ceil(nc/4): nc = 24: 36 instead of 48 (48 real rows fill three tiles exactly,
//         24 complex rows waste a quarter of two), nc = 8: 4 instead of 8.  At 16 systems the kernel is MFMA-bound, so that is its time.
// M16 (with M32): the matrices are stored as complex<half> (NC % 4 == 0): a lane's 16-B load carries four elements, widened to the complex<float>
// tile when they are parked; everything behind the tile is the M32 form.
// (-DQMG_KC_F32_PF2=1: the all-complex<float> MODE 1 form with two pieces of prefetch under a 128-register cap (4 wavefronts per SIMD): nc = 24
// spills 16 registers and goes 1.44 -> 1.61 ms, nc = 12 / 16 within 4 %.  With complex<half> matrices and complex<float> vectors the same
// launch takes 1.21 ms for HALF the matrix bytes: at 8 systems the kernel's floor is its per-piece chain of LDS hand-offs and dependent MFMAs
// (four accumulators), not the stream.)
#ifndef QMG_KC_F32_PF2
#define QMG_KC_F32_PF2 0
#endif
// PAIR (NC = 16, MODE 1, VL): the wavefront owns TWO adjacent nc = 8 sites of a row.  Their 8 x 8 matrices sit on the diagonal of the 16 x 16
// tile (the off-diagonal blocks are zeroed once and never written), their vectors side by side in the 16-wide vector slice -- the two sites'
// own-site and y-neighbour vectors are contiguous in memory, the x-neighbours are found per lane (they wrap at the row ends).  Same MFMA
// count per site as the one-site form (a 16-row tile is half empty at nc = 8 either way), HALF the loads, LDS hand-offs and address
// arithmetic per site: at nc = 8 the one-site form is bound by its instruction issue, not by the stream or the matrix pipe.
template <int NC, int MODE, bool M32, bool V32, bool VL, bool M16 = false, bool PAIR = false>
__global__ __launch_bounds__(BLOCK, (QMG_KC_F32_PF2 && MODE == 1 && M32 && V32 && NC <= 24) ? 4 : (MODE == 1 && NC <= 24) ? 3 : 1) void k_stencil_mfma(const StencilArgs a, const int nk) {
  static_assert(MODE != 2 || (VL && !M32 && NC % 2 == 0), "MODE 2: fp64, right-hand sides through the LDS slice");
  static_assert(!M16 || (M32 && NC % 4 == 0), "16-bit matrices: the fp32 tile path, quads that stay inside a row");
  static_assert(!PAIR || (NC == 16 && MODE == 1 && VL), "PAIR: two nc = 8 sites, packed columns, vectors through the LDS slice");
  constexpr int SNC = PAIR ? NC / 2 : NC;            // colours of ONE site
  constexpr int MEL = PAIR ? 2 * SNC * SNC : NC * NC;   // stored matrix elements per piece per wavefront
  constexpr int RT = (MODE == 2) ? (2 * NC + 15) / 16 : (NC + 15) / 16, KS = (MODE == 2) ? NC / 2 : (NC + 3) / 4;
  constexpr int NACC = (MODE == 2) ? 1 : 2;
  // LDS row stride in tile elements: fp64 tile nc+1 complex (odd: conflict-free 16-B reads); fp32-stored matrices keep the
  // tile as raw complex<float> with stride nc+2 (even: 16-B aligned pair stores) -- half the LDS and half the staging
  // registers, widened to fp64 only as an MFMA operand
  constexpr int RS = M32 ? NC + 2 : NC + 1;
  constexpr int NG = (MEL + WAVE - 1) / WAVE;     // staged 16-B elements per lane per piece
  extern __shared__ __align__(16) unsigned char smem_raw[];
  const int lane = threadIdx.x & (WAVE - 1), wave = threadIdx.x >> 6;
  cplx* mlds = reinterpret_cast<cplx*>(smem_raw) + (size_t)wave * NC * RS;                // fp64 tile
  float2* mlds32 = reinterpret_cast<float2*>(smem_raw) + (size_t)wave * NC * RS;          // fp32 tile (M32)
  const int lr = lane & 15, lq = lane >> 4;
  const int j = (PAIR ? 2 : 1) * (blockIdx.x * (BLOCK / WAVE) + wave);   // (PAIR: the first site of the pair; the host launches it for even hr only)
  if (j >= a.hr) return;                      // whole wavefront leaves; the kernel has no block barriers
  if constexpr (PAIR) {                       // the off-diagonal blocks of the tile: zero for the whole launch
    for (int e = lane; e < NC * RS; e += WAVE) { if (M32) mlds32[e] = make_float2(0.0f, 0.0f); else mlds[e] = cmake(0.0, 0.0); }
    wave_lds_handoff();
  }
  const int kcol = (MODE == 1) ? (lr & 7) : lr;   // right-hand side this lane's MFMA column belongs to
  const bool kval = kcol < nk;                 // ... and whether it exists
  const long koff = (long)system_index(a, kcol & 15) * a.vec_stride;

  for (int row = blockIdx.y; row < a.nrows; row += gridDim.y) {
    const int p = (a.par_count == 2) ? (row & 1) : a.par_first;
    const int y = (a.par_count == 2) ? (row >> 1) : row;
    const bool do_clover = a.clover && ((a.pieces >> p) & 1u);
    const unsigned hop_mask = a.hopping ? ((a.pieces >> (2 + 4 * p)) & 0xFu) : 0u;
    const bool do_shift = (a.pieces >> (10 + p)) & 1u;
    const bool do_zero = (a.pieces >> (12 + p)) & 1u;

    const long site = (long)p * a.half_vol + (long)y * a.hr + j;
    const long opp = (long)(1 - p) * a.half_vol;
    const int s = (y + p) & 1;
    const int yp = (y + 1 == a.Ly) ? 0 : y + 1;
    const int ym = (y == 0) ? a.Ly - 1 : y - 1;
    int jp = j + s; if (jp == a.hr) jp = 0;
    int jm = j + s - 1; if (jm < 0) jm = a.hr - 1;
    // piece slots in the reference's accumulation order: clover, +x, +y, -x, -y
    const long nb[5] = {site, opp + (long)y * a.hr + jp, opp + (long)yp * a.hr + j, opp + (long)y * a.hr + jm, opp + (long)ym * a.hr + j};
    const bool act[5] = {do_clover, (bool)(hop_mask & 1u), (bool)(hop_mask & 2u), (bool)(hop_mask & 4u), (bool)(hop_mask & 8u)};

    // complex<float> matrices AND vectors: the products run on the f32 matrix pipe (v_mfma_f32_16x16x4_f32, twice the f64
    // rate on this part; same A / B / C lane maps as the f64 instruction), accumulating in fp32 like the rest of the fp32 path
    constexpr bool F32M = M32 && V32;
    typedef typename std::conditional<F32M, v4f32, v4d>::type accv;
    accv acc[NACC][RT];   // MODE 0: (re, im); MODE 1: (P, Q)
#pragma unroll
    for (int n = 0; n < NACC; n++)
#pragma unroll
      for (int t = 0; t < RT; t++) {
        if constexpr (F32M) acc[n][t] = (v4f32){0.0f, 0.0f, 0.0f, 0.0f};
        else acc[n][t] = (v4d){0.0, 0.0, 0.0, 0.0};
      }

    constexpr int NGP = M16 ? (MEL / 4 + WAVE - 1) / WAVE : (MEL / 2 + WAVE - 1) / WAVE;   // staged PAIRS (16-bit: QUADS) per lane per piece (narrow-stored matrices)
    // staging registers for the matrix stream (a second set, two pieces of prefetch, was measured SLOWER: 8 rhs 2.88 -> 3.10
    // ms; the registers cost a resident wavefront and the stream was not the limit -- profiles/r02_mfma_kernelC_variants.txt)
    constexpr int NGS = M32 ? NGP : NG;
    // How many pieces of the matrix stream a wavefront keeps in flight (register sets): ONE.  Deeper prefetch was measured for every shape
    // (-DQMG_KC_PFD_A/B/C = sets for <= 1 / <= 3 / <= 5 staged elements per lane; tools/kernelc_bench.py, gpurun_out/r03_kernelc_*.txt): two
    // sets cost fp64 nc = 24 a resident wavefront (round 2: 2.88 -> 3.10 ms); for the half-size fp32-stored stream they fit (124 -> 147 VGPRs)
    // and changed nothing (nc = 24, 8 systems: 1.43 -> 1.54 ms), and five sets at nc = 8 were slower (1.64 -> 1.83 ms): the wavefronts are
    // parked 60 % of their cycles (SQ_WAIT_ANY) with the matrix pipe 37 % busy, but more loads in flight per wavefront do not shorten that.
    // What did: the right-hand sides' system indices without a memory access (system_index) -- a.ridx[k] with a per-lane k is a vector load from
    // the kernel arguments whose result the vector loads' addresses waited for, one more memory latency in front of every piece (fp64 nc = 24,
    // 16 systems: 3.47 -> 2.90-3.00 ms; nc = 16 fp32-stored matrices, 8 systems: 0.905 -> 0.74 ms; nc = 12 fp64: 0.83 -> 0.75 ms).
#ifndef QMG_KC_PFD_A
#define QMG_KC_PFD_A 1
#define QMG_KC_PFD_B 1
#define QMG_KC_PFD_C 1
#endif
    constexpr int PFD = (QMG_KC_F32_PF2 && MODE == 1 && M32 && V32 && NC <= 24 && NGS <= 5) ? 2 : (NGS <= 1) ? QMG_KC_PFD_A : (NGS <= 3) ? QMG_KC_PFD_B : (NGS <= 5) ? QMG_KC_PFD_C : 1;
    cplx G[PFD][NGS];   // M32: raw bits of two complex<float> per entry
    // Right-hand sides.  VL = false (round 1): each lane loads its B-operand entries X_k[4q + lq] straight from global memory --
    // 16 right-hand sides x 64-byte pieces per instruction, 16 cache lines touched per load, 6 loads per piece; going from 4
    // to 8 right-hand sides cost 0.48 ms of a 2.9 ms apply.  VL = true: the piece's nk x NC block is loaded COALESCED
    // (lane-linear over [k][c]: whole 384-byte site vectors), parked in a second LDS slice of the wavefront with rows padded
    // to NC+1 (conflict-free 16-byte fragment reads), and the B fragments are read from there just in time.  The epilogue
    // goes back the same way: results into the slice, then coalesced read-modify-write of the output vectors.
    constexpr int XS = NC + 1;                                   // padded row of the vector slice
    constexpr int XROWS = (MODE == 1) ? 8 : 16;                  // right-hand sides a pass can hold (MODE 1: at most 8)
    constexpr int NXG = (XROWS * NC + WAVE - 1) / WAVE;          // staged vector elements per lane per piece
    constexpr int XPF = (VL && NXG <= 2) ? PFD : 1;               // ... and of the right-hand sides (small blocks only: nc = 8, 12)
    // (the staged right-hand sides stay in their STORAGE form until they are parked: widening a complex<float> entry right after its load made the
    // compiler wait for each load in turn -- load, s_waitcnt vmcnt(0), convert, next load -- BEFORE it issued the piece's matrix loads: three
    // serial memory latencies per piece in the complex<float> forms, none of them overlapped with the MFMAs of the piece in hand)
    typedef typename std::conditional<V32, double, cplx>::type xraw;   // V32: the raw bits of a complex<float>
    xraw XG[XPF][VL ? NXG : 1];
    auto ld_xraw = [](const void* base, long i) -> xraw {       // base == nullptr: a column beyond the systems of the pass (zero)
      if constexpr (V32) return base ? reinterpret_cast<const double*>(base)[i] : 0.0;
      else return base ? reinterpret_cast<const cplx*>(base)[i] : cmake(0.0, 0.0);
    };
    auto widen_xraw = [](xraw v) -> cplx {
      if constexpr (V32) { struct F2 { float x, y; }; const F2 f = __builtin_bit_cast(F2, v); return cmake((double)f.x, (double)f.y); }
      else return v;
    };
    int ksys[VL ? NXG : 1];   // the system each of this lane's staged vector elements belongs to (row-invariant, no memory access: system_index)
    if constexpr (VL) {
#pragma unroll
      for (int g = 0; g < NXG; g++) { const int k = (g * WAVE + lane) / NC; ksys[g] = system_index(a, k < 16 ? k : 0); }
    }
    cplx* xlds = reinterpret_cast<cplx*>(smem_raw + (M32 ? sizeof(float2) : sizeof(cplx)) * (size_t)(BLOCK / WAVE) * NC * RS) + (size_t)wave * XROWS * XS;
    // MODE 1 needs only the half of X its column carries (re for columns 0-7, im for 8-15): one double per k-step
    typename std::conditional<MODE == 1, double, cplx>::type B[2][VL ? 1 : KS];
    auto nb_of = [&](int pc) -> long {            // neighbour site of piece slot pc
      return pc == 0 ? site : pc == 1 ? nb[1] : pc == 2 ? nb[2] : pc == 3 ? nb[3] : nb[4];
    };
    auto load_matrix = [&](int pc, int gs) {      // global -> registers, lane-linear, non-temporal
      const cplx* mbase = (pc == 0) ? a.clover : a.hopping;
      const long moff = ((pc == 0) ? 0 : (long)(pc - 1) * a.size_cm) + site * (SNC * SNC);   // (PAIR: the two sites' matrices are adjacent)
      if (M32) {   // pairs of complex<float> (M16: quads of complex<half>): 16 B per lane per load, kept as raw bits
#pragma unroll
        for (int g = 0; g < NGP; g++) {
          constexpr int PER = M16 ? 4 : 2;
          const int el = PER * (g * WAVE + lane);
          if (MEL % (PER * WAVE) == 0 || el < MEL) {
            const double* pp = M16 ? reinterpret_cast<const double*>(reinterpret_cast<const unsigned*>(mbase) + moff + el)
                                   : reinterpret_cast<const double*>(reinterpret_cast<const float2*>(mbase) + moff + el);
            G[gs][g].x = __builtin_nontemporal_load(pp);
            G[gs][g].y = __builtin_nontemporal_load(pp + 1);
          } else G[gs][g] = cmake(0.0, 0.0);
        }
      } else {
#pragma unroll
        for (int g = 0; g < NG; g++) {
          const int el = g * WAVE + lane;
          G[gs][g] = (MEL % WAVE == 0 || el < MEL) ? ldm<M32, true>(mbase, moff + el) : cmake(0.0, 0.0);
        }
      }
    };
    auto load_vectors = [&](int pc, int set, int xs) {    // the k right-hand sides at the piece's neighbour site (xs: XG register set)
      // a y-slab's rows -1 / Ly: the piece's neighbour row comes from the halo buffer ([system][parity][hr][NC]; a row-uniform choice)
      const bool h_hi = pc == 2 && a.halo_hi && y + 1 == a.Ly, h_lo = pc == 4 && a.halo_lo && y == 0;
      const void* vbase = h_hi ? a.halo_hi : h_lo ? a.halo_lo : a.rhs;
      const long vsite = (h_hi || h_lo) ? (long)(1 - p) * a.hr + j : nb_of(pc);
      const long vstride = (h_hi || h_lo) ? a.halo_stride : a.vec_stride;
      if constexpr (VL && PAIR) {                 // [k][two sites x 8]: the second site's x-neighbour is found per lane (row-end wrap)
        const int sp = (lane & 15) >> 3;          // (NC = 16 divides the wavefront: column = lane % 16 for every g)
        long vs = vsite + sp;                     // own site and y-neighbours: adjacent sites
        if (pc == 1) { int jq = j + sp + s; if (jq >= a.hr) jq -= a.hr; vs = opp + (long)y * a.hr + jq; }
        if (pc == 3) { int jq = j + sp + s - 1; if (jq < 0) jq += a.hr; vs = opp + (long)y * a.hr + jq; }
        const long so = vs * SNC + (lane & 7);
#pragma unroll
        for (int g = 0; g < NXG; g++) {
          const int k = (g * WAVE + lane) / NC;
          XG[xs][g] = ld_xraw((k < nk) ? vbase : nullptr, (long)ksys[g] * vstride + so);
        }
      } else if constexpr (VL) {                  // lane-linear over [k][c]: element e = g*64 + lane -> (k = e / NC, c = e % NC)
        const long so = vsite * NC;
#pragma unroll
        for (int g = 0; g < NXG; g++) {
          const int e = g * WAVE + lane;
          const int k = e / NC, c = e - k * NC;
          XG[xs][g] = ld_xraw((k < nk) ? vbase : nullptr, (long)ksys[g] * vstride + so + c);
        }
      } else {
        const long xo = (long)system_index(a, kval ? (kcol & 15) : 0) * vstride + vsite * NC;    // B-operand layout straight from global memory (a column beyond the pass: system 0's address, value zeroed)
        typename XRaw<V32>::type xr[KS];            // all of them requested before any is widened
        // (unconditional: a lane whose column is beyond the pass reads the pass's first system, a k-step beyond nc reads entry 0; both are zeroed
        // below.  A divergent branch around each load closed it with a full wait.)
#pragma unroll
        for (int q = 0; q < KS; q++) {
          const int c = 4 * q + lq;
          xr[q] = ldv_raw<V32>(vbase, xo + ((NC % 4 == 0 || c < NC) ? c : 0));
        }
#pragma unroll
        for (int q = 0; q < KS; q++) {
          const int c = 4 * q + lq;
          const cplx xw = widen_raw<V32>(xr[q]);
          const cplx xv = (kval && (NC % 4 == 0 || c < NC)) ? xw : cmake(0.0, 0.0);
          if constexpr (MODE == 1) B[set][q] = (lr < 8) ? xv.x : xv.y;
          else B[set][q] = xv;
        }
      }
    };
    auto park_piece = [&](int gs, int xs) {       // registers -> this wavefront's LDS slice, padded rows
      wave_lds_handoff();                         // the previous piece's fragment reads are done
      if (M32) {
#pragma unroll
        for (int g = 0; g < NGP; g++) {
          constexpr int PER = M16 ? 4 : 2;
          const int el = PER * (g * WAVE + lane);
          if (MEL % (PER * WAVE) == 0 || el < MEL) {
            // tile position of stored element el: row-major nc x nc -- PAIR: site sp = el / 64 owns the diagonal block (sp, sp)
            const int trow = PAIR ? (el >> 6) * SNC + ((el & 63) >> 3) : el / NC, tcol = PAIR ? (el >> 6) * SNC + (el & 7) : el % NC;
            if constexpr (M16) {   // (re, im) x 4 halves -> two 16-B stores of complex<float> pairs (RS even, el % 4 == 0: aligned, same row)
              typedef _Float16 h8 __attribute__((ext_vector_type(8)));
              typedef float f4 __attribute__((ext_vector_type(4)));
              const h8 hv = __builtin_bit_cast(h8, G[gs][g]);
              const f4 w0 = {(float)hv[0], (float)hv[1], (float)hv[2], (float)hv[3]}, w1 = {(float)hv[4], (float)hv[5], (float)hv[6], (float)hv[7]};
              float2* dst = mlds32 + trow * RS + tcol;
              *reinterpret_cast<f4*>(dst) = w0;
              *reinterpret_cast<f4*>(dst + 2) = w1;
            } else
              *reinterpret_cast<cplx*>(mlds32 + trow * RS + tcol) = G[gs][g];   // 16-B aligned: RS, el even
          }
        }
      } else {
#pragma unroll
        for (int g = 0; g < NG; g++) {
          const int el = g * WAVE + lane;
          const int trow = PAIR ? (el >> 6) * SNC + ((el & 63) >> 3) : el / NC, tcol = PAIR ? (el >> 6) * SNC + (el & 7) : el % NC;
          if (MEL % WAVE == 0 || el < MEL) mlds[trow * RS + tcol] = G[gs][g];
        }
      }
      if constexpr (VL) {                         // the right-hand sides of the same piece, rows padded
#pragma unroll
        for (int g = 0; g < NXG; g++) {
          const int e = g * WAVE + lane;
          const int k = e / NC, c = e - k * NC;
          if (k < XROWS) xlds[k * XS + c] = widen_xraw(XG[xs][g]);
        }
      }
      wave_lds_handoff();
    };
    auto mac_piece = [&](int set) {
      if constexpr (MODE == 2) {
        // operands of k-step q+1 are read from LDS while the MFMAs of k-step q issue; the scheduling barrier keeps the compiler from
        // hoisting ALL 48 operand reads of the piece in front of the first MFMA (238 VGPRs, one wavefront per SIMD)
        double av[2][RT], bv[2];
        auto fetch = [&](int q, int slot) {
          const int K = 4 * q + lq;                      // 0 .. 2 nc - 1: the first nc multiply Re x, the rest Im x
          const bool khi = K >= NC;
          const int kc = khi ? K - NC : K;
          bv[slot] = reinterpret_cast<const double*>(xlds + kcol * XS + kc)[khi ? 1 : 0];
#pragma unroll
          for (int t = 0; t < RT; t++) {
            const int R = 16 * t + lr;                   // 0 .. 2 nc - 1: the first nc are Re y, the rest Im y
            const bool rhi = R >= NC;
            const int rr = rhi ? R - NC : R;
            double v = 0.0;
            if ((2 * NC) % 16 == 0 || R < 2 * NC) {
              v = reinterpret_cast<const double*>(mlds + rr * RS + kc)[rhi != khi ? 1 : 0];   // diagonal quadrants: Re M; off-diagonal: Im M ...
              if (!rhi && khi) v = -v;                                                        // ... with a minus in the upper right one
            }
            av[slot][t] = v;
          }
        };
        fetch(0, 0);
#pragma unroll
        for (int q = 0; q < KS; q++) {
          if (q + 1 < KS) fetch(q + 1, (q + 1) & 1);
#pragma unroll
          for (int t = 0; t < RT; t++) acc[0][t] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[q & 1][t], bv[q & 1], acc[0][t], 0, 0, 0);
          __builtin_amdgcn_sched_barrier(0);
        }
      } else {
#pragma unroll
      for (int q = 0; q < KS; q++) {
        cplx Af[RT];
#pragma unroll
        for (int t = 0; t < RT; t++) {
          const int r = 16 * t + lr, c = 4 * q + lq;
          if (M32) {
            const float2 mf = ((NC % 16 == 0 || r < NC) && (NC % 4 == 0 || c < NC)) ? mlds32[r * RS + c] : make_float2(0.0f, 0.0f);
            Af[t] = cmake((double)mf.x, (double)mf.y);
          } else
            Af[t] = ((NC % 16 == 0 || r < NC) && (NC % 4 == 0 || c < NC)) ? mlds[r * RS + c] : cmake(0.0, 0.0);
        }
        if constexpr (VL) {                       // B fragment of this k-step: X_{column}[4q + lq] from the wavefront's vector slice
          const int c = 4 * q + lq;
          const cplx xv = (NC % 4 == 0 || c < NC) ? xlds[kcol * XS + c] : cmake(0.0, 0.0);
          if constexpr (MODE == 1) B[set][0] = (lr < 8) ? xv.x : xv.y;
          else B[set][0] = xv;
        }
        constexpr int qb = VL ? 0 : 1;            // VL: the fragment sits in slot 0; else slot q
        if constexpr (F32M && MODE == 0) {
#pragma unroll
          for (int t = 0; t < RT; t++) {
            acc[0][t] = __builtin_amdgcn_mfma_f32_16x16x4f32((float)Af[t].x, (float)B[set][q * qb].x, acc[0][t], 0, 0, 0);
            acc[1][t] = __builtin_amdgcn_mfma_f32_16x16x4f32((float)Af[t].x, (float)B[set][q * qb].y, acc[1][t], 0, 0, 0);
          }
#pragma unroll
          for (int t = 0; t < RT; t++) {
            acc[0][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(-(float)Af[t].y, (float)B[set][q * qb].y, acc[0][t], 0, 0, 0);
            acc[1][t] = __builtin_amdgcn_mfma_f32_16x16x4f32((float)Af[t].y, (float)B[set][q * qb].x, acc[1][t], 0, 0, 0);
          }
        } else if constexpr (F32M) {
#pragma unroll
          for (int t = 0; t < RT; t++) {
            acc[0][t] = __builtin_amdgcn_mfma_f32_16x16x4f32((float)Af[t].x, (float)B[set][q * qb], acc[0][t], 0, 0, 0);
            acc[1][t] = __builtin_amdgcn_mfma_f32_16x16x4f32((float)Af[t].y, (float)B[set][q * qb], acc[1][t], 0, 0, 0);
          }
        } else if constexpr (MODE == 0) {
#pragma unroll
          for (int t = 0; t < RT; t++) {
            acc[0][t] = __builtin_amdgcn_mfma_f64_16x16x4f64(Af[t].x, B[set][q * qb].x, acc[0][t], 0, 0, 0);
            acc[1][t] = __builtin_amdgcn_mfma_f64_16x16x4f64(Af[t].x, B[set][q * qb].y, acc[1][t], 0, 0, 0);
          }
#pragma unroll
          for (int t = 0; t < RT; t++) {
            acc[0][t] = __builtin_amdgcn_mfma_f64_16x16x4f64(-Af[t].y, B[set][q * qb].y, acc[0][t], 0, 0, 0);
            acc[1][t] = __builtin_amdgcn_mfma_f64_16x16x4f64(Af[t].y, B[set][q * qb].x, acc[1][t], 0, 0, 0);
          }
        } else {
#pragma unroll
          for (int t = 0; t < RT; t++) {
            acc[0][t] = __builtin_amdgcn_mfma_f64_16x16x4f64(Af[t].x, B[set][q * qb], acc[0][t], 0, 0, 0);
            acc[1][t] = __builtin_amdgcn_mfma_f64_16x16x4f64(Af[t].y, B[set][q * qb], acc[1][t], 0, 0, 0);
          }
        }
      }
      }
    };

    // software pipeline over the ACTIVE piece slots (activity is uniform over the block): PFD pieces of the matrix stream (XPF of the
    // right-hand sides) are requested ahead of the piece that computes.  Register-set indices are compile-time constants after unrolling;
    // requests are issued oldest-needed-first, so the wait in front of a park leaves the younger ones in flight.
    unsigned am = (act[0] ? 1u : 0u) | (act[1] ? 2u : 0u) | (act[2] ? 4u : 0u) | (act[3] ? 8u : 0u) | (act[4] ? 16u : 0u);
    const int n = __popc(am);
    int lst[5];   // the active slots in order (constant indices only: stays in registers)
#pragma unroll
    for (int i = 0; i < 5; i++) { lst[i] = am ? __ffs(am) - 1 : 0; am &= am - 1; }
#pragma unroll
    for (int i = 0; i < PFD; i++)
      if (i < n) {
        load_matrix(lst[i], i);
        if (i == 0 || XPF > 1) load_vectors(lst[i], i & 1, XPF > 1 ? i : 0);
      }
#pragma unroll
    for (int i = 0; i < 5; i++) {
      if (i < n) {
        park_piece(i % PFD, XPF > 1 ? i % PFD : 0);   // the sets that held piece i are free again after this
        if (XPF == 1 && i + 1 < n) load_vectors(lst[i + 1], (i + 1) & 1, 0);
        if (i + PFD < n) {
          load_matrix(lst[i + PFD], i % PFD);
          if (XPF > 1) load_vectors(lst[i + PFD], (i + PFD) & 1, i % PFD);
        }
        mac_piece(i & 1);
      }
    }

    // epilogue: shift, accumulate, store.  Lane (lq, lr) owns rows 16 t + 4 i + lq of right-hand side lr.
    const double sg = p ? -1.0 : 1.0;
    if constexpr (VL) {
      // results into the vector slice [k][r] (the last piece's fragment reads are done), then lane-linear over [k][r]:
      // coalesced own-site read for the shift term, coalesced read-modify-write of the output
      wave_lds_handoff();
      if constexpr (MODE == 2) {   // real row R of system lr: Re (R < nc) or Im of output row R mod nc
#pragma unroll
        for (int t = 0; t < RT; t++) {
#pragma unroll
          for (int i = 0; i < 4; i++) {
            const int R = 16 * t + 4 * i + lq;
            if ((2 * NC) % 16 == 0 || R < 2 * NC) reinterpret_cast<double*>(xlds + kcol * XS + (R >= NC ? R - NC : R))[R >= NC ? 1 : 0] = acc[0][t][i];
          }
        }
      } else {
#pragma unroll
      for (int t = 0; t < RT; t++) {
#pragma unroll
        for (int i = 0; i < 4; i++) {
          const int r = F32M ? 16 * t + 4 * lq + i : 16 * t + 4 * i + lq;   // C/D row of accumulator register i: the f32 instruction puts rows 4 lq .. 4 lq + 3 in a lane, the f64 one rows lq, lq + 4, ...
          cplx v;
          if (MODE == 0) v = cmake((double)acc[0][t][i], (double)acc[NACC - 1][t][i]);
          else {   // partner lane (lr ^ 8) holds the other half of the packed columns
            const double pp = (double)__shfl_xor(acc[0][t][i], 8), qp = (double)__shfl_xor(acc[NACC - 1][t][i], 8);
            v = cmake((double)acc[0][t][i] - qp, pp + (double)acc[NACC - 1][t][i]);
          }
          if (r < NC && (MODE != 1 || lr < 8)) xlds[kcol * XS + r] = v;
        }
      }
      }
      wave_lds_handoff();
#pragma unroll
      for (int g = 0; g < NXG; g++) {
        const int e = g * WAVE + lane;
        const int k = e / NC, r = e - k * NC;
        if (k < nk) {
          cplx v = xlds[k * XS + r];
          const long o = (long)ksys[g] * a.vec_stride + site * SNC + r;   // (PAIR: the second site's vector follows the first's)
          if (do_shift) {
            const double dg = (SNC % 2 == 0) ? (((PAIR ? (r & (SNC - 1)) : r) < SNC / 2) ? 1.0 : -1.0) : 0.0;
            const cplx sh = cmake(a.shift[0] + sg * a.eo_shift[0] + dg * a.dof_shift[0], a.shift[1] + sg * a.eo_shift[1] + dg * a.dof_shift[1]);
            cmac(v, sh, ldv<V32>(a.rhs, o));
          }
          if (!do_zero) v = cadd(ldv<V32>(a.lhs, o), v);
          stv<V32>(a.lhs, o, v);
        }
      }
      wave_lds_handoff();   // the next row's first park must not overtake these reads
    } else {
#pragma unroll
      for (int t = 0; t < RT; t++) {
#pragma unroll
        for (int i = 0; i < 4; i++) {
          const int r = F32M ? 16 * t + 4 * lq + i : 16 * t + 4 * i + lq;   // C/D row of accumulator register i: the f32 instruction puts rows 4 lq .. 4 lq + 3 in a lane, the f64 one rows lq, lq + 4, ...
          cplx v;
          if (MODE == 0) v = cmake((double)acc[0][t][i], (double)acc[NACC - 1][t][i]);
          else {   // partner lane (lr ^ 8) holds the other half of the packed columns
            const double pp = (double)__shfl_xor(acc[0][t][i], 8), qp = (double)__shfl_xor(acc[NACC - 1][t][i], 8);
            v = cmake((double)acc[0][t][i] - qp, pp + (double)acc[NACC - 1][t][i]);
          }
          if (r < NC && kval && (MODE != 1 || lr < 8)) {
            const long o = koff + site * NC + r;
            if (do_shift) {
              const double dg = (NC % 2 == 0) ? ((r < NC / 2) ? 1.0 : -1.0) : 0.0;
              const cplx sh = cmake(a.shift[0] + sg * a.eo_shift[0] + dg * a.dof_shift[0], a.shift[1] + sg * a.eo_shift[1] + dg * a.dof_shift[1]);
              cmac(v, sh, ldv<V32>(a.rhs, o));
            }
            if (!do_zero) v = cadd(ldv<V32>(a.lhs, o), v);
            stv<V32>(a.lhs, o, v);
          }
        }
      }
    }
  }
}

static int g_stencil_nt = 3;     // tuning knob: bit0 non-temporal matrix loads, bit1 non-temporal stores (kernel A)
#ifdef QMG_DIAGNOSTICS
static int g_stencil_ablate = 0;
#endif
static int g_stencil_site = 3;    // tuning knob: nc 2 through the site kernel (qmg_site.hip): bit 0 fp64 where it is faster, bit 1 fp32, bit 2 fp64 always
static int g_stencil_pair = 2;    // tuning knob: 0 = one site per lane group (kernel A), 1/2 = paired parities x 1/2 rows (kernel A2)
static int g_pair_prefetch = 1;   // tuning knob: 1 = kernel A2 prefetches the next system's right-hand side in fp64 batches
static int g_stencil_rows = 0;   // tuning knob: cap on gridDim.y (0 = one block row per lattice row)
static int g_stencil_mfma = 1;   // tuning knob: 1 = multi-rhs applies with nc in {8,12,16,24,32} run on the f64 matrix cores (kernel C); 2 = same, plain 4-MFMA products; 0 = off
static int g_mfma_pair8 = 1;     // tuning knob: 1 = kernel C at nc = 8 with up to 8 systems owns two sites per wavefront, 0 = one
static int g_mfma_vl = 1;        // tuning knob: 1 = kernel C loads / stores the right-hand sides coalesced through an LDS slice, 0 = operand-layout global accesses
static int g_gen32 = 1;          // tuning knob: fp32-stored matrices, even nc: 1 = kernel B32 (fp32 tile end to end), 2 = same with 2-site tiles, 0 = kernel B with widening loads
static int g_gen_sites = 0;      // tuning knob: cap on sites per block in kernel B (0 = register-limited maximum)

static GenLayout make_gen_layout(int nc, int hr, int mat32, int site_cap = 0) {
  GenLayout L;
  const int nc2 = nc * nc;
  int S = (BLOCK * GEN_MAX_PER_THREAD) / nc2;       // registers: S*nc^2 <= 256*12
  if (S > BLOCK / nc) S = BLOCK / nc;               // one (s,r) row per thread at least
  if (S > hr) S = hr;
  // fp32-stored matrices: the kernel is bound by bytes in flight per CU (one piece per resident block), not by HBM; with
  // half the bytes per piece, smaller tiles (more resident blocks) pay: 512^2, nc = 24: S = 5 1.81 ms, S = 2 1.59 ms
  if (mat32 && nc >= 16 && S > 2) S = 2;
  if (g_gen_sites > 0 && S > g_gen_sites) S = g_gen_sites;
  if (site_cap > 0 && S > site_cap) S = site_cap;
  if (S < 1) S = 1;
  L.S = S;
  int H = BLOCK / (S * nc);
  if (H < 1) H = 1;
  if (H > nc) H = nc;
  L.H = H;
  L.rs = nc + ((nc % 2 == 0) ? 1 : 0);
  L.mat_elems = S * nc2;
  L.per_thread = (L.mat_elems + BLOCK - 1) / BLOCK;
  return L;
}

}  // namespace qmg

using namespace qmg;

extern "C" int qmg_set_tuning(const char* key, int value) {
  if (!key) return QMG_ERR_INVALID;
  if (!strcmp(key, "stencil_nt")) { g_stencil_nt = value; return QMG_SUCCESS; }
#ifdef QMG_DIAGNOSTICS
  if (!strcmp(key, "stencil_ablate")) { g_stencil_ablate = value; return QMG_SUCCESS; }
#endif
  if (!strcmp(key, "stencil_pair")) { g_stencil_pair = value; return QMG_SUCCESS; }
  if (!strcmp(key, "blas_nt_mb")) { g_blas_nt_bytes = (long)value << 20; return QMG_SUCCESS; }
  if (!strcmp(key, "pair_prefetch")) { g_pair_prefetch = value; return QMG_SUCCESS; }
  if (!strcmp(key, "stencil_site")) { g_stencil_site = value; return QMG_SUCCESS; }
  if (!strcmp(key, "site_block")) { if (value != 64 && value != 128 && value != 256) return QMG_ERR_INVALID; g_site_block = value; return QMG_SUCCESS; }
  if (!strcmp(key, "site_gy")) { g_site_gy = value; return QMG_SUCCESS; }
  if (!strcmp(key, "site_generic")) { g_site_generic = value ? 1 : 0; return QMG_SUCCESS; }
  if (!strcmp(key, "stencil_rows")) { g_stencil_rows = value; return QMG_SUCCESS; }
  if (!strcmp(key, "gen_sites")) { g_gen_sites = value; return QMG_SUCCESS; }
  if (!strcmp(key, "gen32")) { g_gen32 = value; return QMG_SUCCESS; }
  if (!strcmp(key, "stencil_mfma")) { g_stencil_mfma = value; return QMG_SUCCESS; }
  if (!strcmp(key, "mfma_vl")) { g_mfma_vl = value; return QMG_SUCCESS; }
  if (!strcmp(key, "mfma_pair8")) { g_mfma_pair8 = value; return QMG_SUCCESS; }
  if (!strcmp(key, "xfer_tile")) { g_xfer_tile = value; return QMG_SUCCESS; }
  if (!strcmp(key, "xfer_pack")) { g_xfer_pack = value; return QMG_SUCCESS; }
  if (!strcmp(key, "wilson_pair")) { g_wilson_pair = value; return QMG_SUCCESS; }
  if (!strcmp(key, "setup_fused")) { g_setup_fused = value; return QMG_SUCCESS; }
  if (!strcmp(key, "xfer_mfma")) { g_xfer_mfma = value; return QMG_SUCCESS; }
  if (!strcmp(key, "reduce_spin")) { g_reduce_spin = value; return QMG_SUCCESS; }
  if (!strcmp(key, "malloc_poison")) { g_malloc_poison = value ? 1 : 0; return QMG_SUCCESS; }
  return QMG_ERR_INVALID;
}

static int stencil_apply_impl(const qmg_stencil_desc* d, void* lhs, const void* rhs, unsigned pieces, int nrhs, size_t vec_stride,
                              const unsigned char* ridx, void* stream, int mat32 = 0, int vec32 = 0, const SlabHalo* slab = nullptr,
                              double* norms_dev = nullptr, const qmg_apply_epilogue* epi = nullptr);

// One system with an epilogue on the finished site values (include/qmg_hip.h: qmg_apply_epilogue).  dtype QMG_C64: fp64 matrices and
// vectors; mat32 == 1: complex<float> matrices (d->clover / d->hopping point to float pairs), mat32 == 2: complex<half> matrices, with fp64 vectors;
// QMG_C32: fp32 vectors with either.
// QMG_ERR_UNSUPPORTED where the dispatch lands on a kernel without the epilogue (nc = 1, 2, 4; batches): the caller runs the
// separate passes instead.
extern "C" int qmg_stencil_apply_epi_t(int dtype, int mat32, const qmg_stencil_desc* d, void* lhs, const void* rhs, unsigned pieces, size_t vec_stride, int system,
                                       const qmg_apply_epilogue* epi, void* stream) {
  if (!epi || (dtype != QMG_C64 && dtype != QMG_C32) || system < 0 || system > 15) return QMG_ERR_INVALID;
  if (dtype == QMG_C32 && !mat32) return QMG_ERR_INVALID;
  if (mat32 && d && (d->nc == 1 || d->nc == 2 || d->nc == 4)) return QMG_ERR_UNSUPPORTED;
  unsigned char ridx[16];
  for (int k = 0; k < 16; k++) ridx[k] = (unsigned char)system;
  return stencil_apply_impl(d, lhs, rhs, pieces, 1, vec_stride, system ? ridx : nullptr, stream, mat32 == 2 ? 2 : mat32 ? 1 : 0, dtype == QMG_C32 ? 1 : 0, nullptr, nullptr, epi);
}

// partials of the fused norms (one buffer per host thread = per rank, grown on demand) and the default result slot
// part: the fused-norm partials of the calling thread.  One buffer per thread, so two calls of one thread on DIFFERENT streams would race on it:
// `done` is recorded behind every use and a call on another stream than the last one waits for it first (same stream: stream order suffices).
struct NormWorkspace { double* part = nullptr; size_t cap = 0; int device = -1; double* own = nullptr; int own_dev = -1;
                       hipEvent_t done = nullptr; hipStream_t last = nullptr; bool used = false; };
static thread_local NormWorkspace g_norm_ws;
namespace qmg {
void release_stencil_workspace() {   // qmg_shutdown (qmg_runtime.hip)
  if (g_norm_ws.part) (void)hipFree(g_norm_ws.part);
  if (g_norm_ws.own) (void)hipFree(g_norm_ws.own);
  if (g_norm_ws.done) (void)hipEventDestroy(g_norm_ws.done);
  g_norm_ws = NormWorkspace();
}
}  // namespace qmg

extern "C" int qmg_stencil_apply(const qmg_stencil_desc* d, void* lhs, const void* rhs, unsigned pieces,
                                 int nrhs, size_t vec_stride, void* stream) {
  return stencil_apply_impl(d, lhs, rhs, pieces, nrhs, vec_stride, nullptr, stream);
}

// Masked batch: only the right-hand sides whose bit is set in `mask` are read or written (a lock-step batched solver
// freezes the systems that have converged).  At most 16 right-hand sides per call.
extern "C" int qmg_stencil_apply_batch(const qmg_stencil_desc* d, void* lhs, const void* rhs, unsigned pieces,
                                       int nrhs, size_t vec_stride, unsigned mask, void* stream) {
  if (nrhs < 1 || nrhs > 16) return QMG_ERR_INVALID;
  unsigned char ridx[16];
  int n = 0;
  for (int k = 0; k < nrhs; k++)
    if ((mask >> k) & 1u) ridx[n++] = (unsigned char)k;
  if (n == 0) return QMG_SUCCESS;
  if (n == nrhs) return stencil_apply_impl(d, lhs, rhs, pieces, nrhs, vec_stride, nullptr, stream);
  return stencil_apply_impl(d, lhs, rhs, pieces, n, vec_stride, ridx, stream);
}

// Matrices stored as complex<float> (d->clover / d->hopping point to float pairs), everything else fp64: vectors, shifts,
// accumulation.  Halves the matrix stream of the HBM-bound coarse applies.  An OPT-IN storage format for operators that
// only precondition (the K-cycle inside a flexible fp64 outer solver); nc = 1, 2, 4 are not served.
extern "C" int qmg_stencil_apply_mat32(const qmg_stencil_desc* d, void* lhs, const void* rhs, unsigned pieces,
                                       int nrhs, size_t vec_stride, unsigned mask, void* stream) {
  if (!d || d->nc == 1 || d->nc == 2 || d->nc == 4) return QMG_ERR_UNSUPPORTED;
  if (nrhs < 1 || nrhs > 16) return QMG_ERR_INVALID;
  unsigned char ridx[16];
  int n = 0;
  for (int k = 0; k < nrhs; k++)
    if ((mask >> k) & 1u) ridx[n++] = (unsigned char)k;
  if (n == 0) return QMG_SUCCESS;
  if (n == nrhs) return stencil_apply_impl(d, lhs, rhs, pieces, nrhs, vec_stride, nullptr, stream, 1);
  return stencil_apply_impl(d, lhs, rhs, pieces, n, vec_stride, ridx, stream, 1);
}

// Matrices stored as complex<half> (d->clover / d->hopping point to __half2 pairs: qmg_convert_to_c16), vectors complex<double> (QMG_C64) or
// complex<float> (QMG_C32), accumulation fp64.  A quarter of the fp64 matrix stream.  For operators that only PRECONDITION; nc a multiple of 4
// (the Galerkin operators: 8, 12, 16, 24, 32); QMG_ERR_UNSUPPORTED otherwise.  The values must be inside half range (|x| < 65504; magnitudes
// below 6e-8 flush to zero): the caller checks that when it converts.
extern "C" int qmg_stencil_apply_mat16_t(int vec_dtype, const qmg_stencil_desc* d, void* lhs, const void* rhs, unsigned pieces,
                                         int nrhs, size_t vec_stride, unsigned mask, void* stream) {
  if (vec_dtype != QMG_C64 && vec_dtype != QMG_C32) return QMG_ERR_INVALID;
  if (!d || (d->nc & 3) || d->nc == 4) return QMG_ERR_UNSUPPORTED;
  if (nrhs < 1 || nrhs > 16) return QMG_ERR_INVALID;
  unsigned char ridx[16];
  int n = 0;
  for (int k = 0; k < nrhs; k++)
    if ((mask >> k) & 1u) ridx[n++] = (unsigned char)k;
  if (n == 0) return QMG_SUCCESS;
  return stencil_apply_impl(d, lhs, rhs, pieces, n == nrhs ? nrhs : n, vec_stride, n == nrhs ? nullptr : ridx, stream, 2, vec_dtype == QMG_C32 ? 1 : 0);
}

// lhs_k (+)= pieces(M) rhs_k and norms[k] = |lhs_k|^2 from the same pass (the vector is not read again): fp64, nc = 1 or 2,
// both parities written, lhs != rhs, nrhs <= 16 (QMG_ERR_UNSUPPORTED otherwise; also under distributed reductions, where the
// caller sums the norms itself).  norms_dev: nrhs doubles in device memory, or NULL; norms_host: nrhs doubles, or NULL
// (synchronises the stream).  The bytes of lhs are those qmg_stencil_apply writes.
extern "C" int qmg_stencil_apply_norm2(const qmg_stencil_desc* d, void* lhs, const void* rhs, unsigned pieces, int nrhs, size_t vec_stride,
                                       double* norms_dev, double* norms_host, void* stream) {
  if (!norms_dev && !norms_host) return QMG_ERR_INVALID;
  if (nrhs < 1 || nrhs > 16) return QMG_ERR_INVALID;
  if (dist_reductions_on()) return QMG_ERR_UNSUPPORTED;
  if (!(pieces & (QMG_P_CLOVER_E | QMG_P_EO | QMG_P_SHIFT_E | QMG_P_ZERO_E)) || !(pieces & (QMG_P_CLOVER_O | QMG_P_OE | QMG_P_SHIFT_O | QMG_P_ZERO_O)))
    return QMG_ERR_UNSUPPORTED;   // a parity left untouched: its part of |lhs|^2 is not seen by the kernel
  double* res = norms_dev;
  if (!res) {
    int dev = 0;
    QMG_HIP_CHECK(hipGetDevice(&dev));
    NormWorkspace& ws = g_norm_ws;
    if (ws.own_dev != dev) { QMG_HIP_CHECK(hipMalloc((void**)&ws.own, sizeof(double) * 16)); ws.own_dev = dev; }
    res = ws.own;
  }
  const int rc = stencil_apply_impl(d, lhs, rhs, pieces, nrhs, vec_stride, nullptr, stream, 0, 0, nullptr, res);
  if (rc) return rc;
  if (norms_host) {
    QMG_HIP_CHECK(hipMemcpyAsync(norms_host, res, sizeof(double) * nrhs, hipMemcpyDeviceToHost, as_stream(stream)));
    QMG_HIP_CHECK(hipStreamSynchronize(as_stream(stream)));
  }
  return QMG_SUCCESS;
}

// Either storage precision, masked batch semantics.  QMG_C64: qmg_stencil_apply_batch.  QMG_C32: matrices AND vectors are
// complex<float>; nc in {1,2,4} run kernel A in fp32 arithmetic, every other nc the fp32-tile kernels B32 / B / C with
// fp32 vector loads and stores around their fp64 accumulation.
extern "C" int qmg_stencil_apply_t(int dtype, const qmg_stencil_desc* d, void* lhs, const void* rhs, unsigned pieces,
                                   int nrhs, size_t vec_stride, unsigned mask, void* stream) {
  if (dtype == QMG_C64) return qmg_stencil_apply_batch(d, lhs, rhs, pieces, nrhs, vec_stride, mask, stream);
  if (dtype != QMG_C32) return QMG_ERR_INVALID;
  if (nrhs < 1 || nrhs > 16) return QMG_ERR_INVALID;
  unsigned char ridx[16];
  int n = 0;
  for (int k = 0; k < nrhs; k++)
    if ((mask >> k) & 1u) ridx[n++] = (unsigned char)k;
  if (n == 0) return QMG_SUCCESS;
  if (n == nrhs) return stencil_apply_impl(d, lhs, rhs, pieces, nrhs, vec_stride, nullptr, stream, 1, 1);
  return stencil_apply_impl(d, lhs, rhs, pieces, n, vec_stride, ridx, stream, 1, 1);
}

// the epilogue's dot partials: one slot per wavefront of the launch (system slot 0), summed by mr_epilogue_finish into the thread's MR slot
#define QMG_EPI_BEGIN(GX, GY)                                                                   \
  long epi_npart = 0;                                                                           \
  if (a.epi.on && a.epi.dotv) {                                                                 \
    const unsigned epi_cap = (GX) >= 2048u ? 1u : 2048u / (GX);   /* a few thousand partials for the one-block second stage: blocks walk rows */ \
    if (grid.y > epi_cap) grid.y = epi_cap;                                                     \
    epi_npart = (long)(GX) * (long)grid.y * (BLOCK / WAVE);                                       \
    a.epi.part = mr_epilogue_begin(1, epi_npart);                                               \
    a.epi.npart = epi_npart;                                                                    \
    if (!a.epi.part) return QMG_ERR_HIP;                                                        \
  }
#define QMG_EPI_FINISH()                                                                        \
  if (epi_npart) { const unsigned char id0 = a.ridx[0]; const int erc = mr_epilogue_finish(&id0, 1, epi_npart, st); if (erc) return erc; }

// The 1 x 1 lattice (lattice.h:77,201; stencil_2d.h:870-888, "this corner case is annoying").  Every half-volume loop of the reference runs
// volume / 2 = 0 times there -- the clover sweeps, the cshifts and the hopping cMATxpy's touch nothing -- so apply_M is its shift term alone:
// the one site counts as even, lhs[c] += (shift + eo_shift +- dof_shift) rhs[c] (dof_shift only for even nc, + on the first half).
// The zero pieces clear the site.  One tiny launch; plain applies only.
template <typename T>
__global__ void k_stencil_volume1(void* lhs_, const void* rhs_, int nc, int nrhs, long stride, const unsigned char* ridx_dev_unused, StencilArgs a, int zero, int shift_on) {
  typedef typename CStore<T>::type ct;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nc * nrhs) return;
  const int k = i / nc, c = i - k * nc;
  const long o = (long)system_index(a, k) * stride + c;
  ct* lhs = reinterpret_cast<ct*>(lhs_);
  const ct* rhs = reinterpret_cast<const ct*>(rhs_);
  cplx v = cmake(0.0, 0.0);
  if (!zero) { const ct l = lhs[o]; v = cmake((double)l.x, (double)l.y); }
  if (shift_on) {
    const double dg = (nc % 2 == 0) ? ((c < nc / 2) ? 1.0 : -1.0) : 0.0;
    const cplx sh = cmake(a.shift[0] + a.eo_shift[0] + dg * a.dof_shift[0], a.shift[1] + a.eo_shift[1] + dg * a.dof_shift[1]);
    const ct r = rhs[o];
    cmac(v, sh, cmake((double)r.x, (double)r.y));
  }
  ct w; w.x = (T)v.x; w.y = (T)v.y;
  lhs[o] = w;
}

static int stencil_apply_volume1(const qmg_stencil_desc* d, void* lhs, const void* rhs, unsigned pieces, int nrhs, size_t vec_stride,
                                 const unsigned char* ridx, void* stream, int vec32) {
  if (d->nc < 1 || nrhs > 16 || (nrhs > 1 && vec_stride < (size_t)d->nc)) return QMG_ERR_INVALID;
  StencilArgs a;
  memset(&a, 0, sizeof(a));
  a.use_idx = ridx ? 1 : 0;
  for (int k = 0; k < 16; k++) a.ridx[k] = ridx ? ridx[k < nrhs ? k : 0] : (unsigned char)k;
  for (int i = 0; i < 2; i++) { a.shift[i] = d->shift[i]; a.eo_shift[i] = d->eo_shift[i]; a.dof_shift[i] = d->dof_shift[i]; }
  const int zero = (pieces & (QMG_P_ZERO_E | QMG_P_ZERO_O)) ? 1 : 0, shift_on = (pieces & QMG_P_SHIFT_E) ? 1 : 0;
  if (!zero && !shift_on) return QMG_SUCCESS;
  const int n = d->nc * nrhs;
  if (vec32) k_stencil_volume1<float><<<(n + 63) / 64, 64, 0, as_stream(stream)>>>(lhs, rhs, d->nc, nrhs, (long)vec_stride, nullptr, a, zero, shift_on);
  else k_stencil_volume1<double><<<(n + 63) / 64, 64, 0, as_stream(stream)>>>(lhs, rhs, d->nc, nrhs, (long)vec_stride, nullptr, a, zero, shift_on);
  QMG_LAUNCH_CHECK();
  return QMG_SUCCESS;
}

static int stencil_apply_impl(const qmg_stencil_desc* d, void* lhs, const void* rhs, unsigned pieces, int nrhs, size_t vec_stride,
                              const unsigned char* ridx, void* stream, int mat32, int vec32, const SlabHalo* slab, double* norms_dev, const qmg_apply_epilogue* epi) {
  if (!d || !lhs || !rhs || nrhs < 1) return QMG_ERR_INVALID;
  if (d->Lx == 1 && d->Ly == 1) {
    if (slab || norms_dev || epi) return QMG_ERR_UNSUPPORTED;
    return stencil_apply_volume1(d, lhs, rhs, pieces, nrhs, vec_stride, ridx, stream, vec32);
  }
  if (!valid_lattice(d->Lx, d->Ly) || d->nc < 1) return QMG_ERR_INVALID;
  const int nc = d->nc;
  if (nrhs > 1 && vec_stride < (size_t)d->Lx * d->Ly * nc) return QMG_ERR_INVALID;

  StencilArgs a;
  a.clover = (const cplx*)d->clover;
  a.hopping = (const cplx*)d->hopping;
  a.lhs = lhs;
  a.rhs = rhs;
  a.vec32 = vec32;
  if (vec32 && !mat32) return QMG_ERR_UNSUPPORTED;   // fp32 vectors come with fp32 matrices (qmg_stencil_apply_t)
  // nc = 2 in one storage precision: the site kernel (kernel S, qmg_site.hip)
  if (slab && nc == 2 && mat32 != vec32) return QMG_ERR_UNSUPPORTED;   // slabs at nc = 2: kernel S, matrices and vectors in ONE precision (or its own 16-bit form)
  if (epi && (norms_dev || nrhs != 1)) return QMG_ERR_UNSUPPORTED;   // the epilogue is served for ONE system per launch, by kernels B / B32
  if (nc == 2 && mat32 == vec32 && nrhs <= 16 && !norms_dev && !epi && (slab || (vec32 ? (g_stencil_site & 2) : (g_stencil_site & 5)))) {
    const int rc = site_kernel_apply(vec32 ? 1 : 2, d, lhs, rhs, pieces, nrhs, (long)vec_stride, ridx, as_stream(stream), !slab && !(g_stencil_site & 4), slab);
    if (rc != SITE_DECLINED) return rc;
  }
  a.hr = d->Lx / 2;
  a.Ly = d->Ly;
  a.half_vol = (long)a.hr * d->Ly;
  a.size_cm = 2 * a.half_vol * nc * nc;
  a.pieces = pieces;
  a.nrhs = nrhs;
  a.vec_stride = (long)vec_stride;
  a.use_idx = ridx ? 1 : 0;
  a.mat32 = mat32 ? 1 : 0;
  a.mat16 = (mat32 == 2) ? 1 : 0;   // (mat32 == 2: complex<half> storage)
  a.halo_lo = slab ? slab->lo : nullptr;
  a.halo_hi = slab ? slab->hi : nullptr;
  a.halo_stride = slab ? slab->stride : 0;
  a.norm_part = nullptr;
  a.epi = no_epilogue();
  for (int k = 0; k < 16; k++) a.ridx[k] = ridx ? ridx[k < nrhs ? k : 0] : (unsigned char)k;
#ifdef QMG_DIAGNOSTICS
  a.ablate = g_stencil_ablate;
#endif
  for (int i = 0; i < 2; i++) { a.shift[i] = d->shift[i]; a.eo_shift[i] = d->eo_shift[i]; a.dof_shift[i] = d->dof_shift[i]; }

  // which parity halves have any work
  const unsigned even_bits = QMG_P_CLOVER_E | QMG_P_EO | QMG_P_SHIFT_E | QMG_P_ZERO_E;
  const unsigned odd_bits = QMG_P_CLOVER_O | QMG_P_OE | QMG_P_SHIFT_O | QMG_P_ZERO_O;
  const bool ev = pieces & even_bits, od = pieces & odd_bits;
  if (!ev && !od) return QMG_SUCCESS;
  a.par_first = ev ? 0 : 1;
  a.par_count = (ev && od) ? 2 : 1;
  a.nrows = d->Ly * a.par_count;
  unsigned gy = a.nrows > 65535 ? 65535u : (unsigned)a.nrows;
  if (g_stencil_rows > 0 && gy > (unsigned)g_stencil_rows) gy = (unsigned)g_stencil_rows;
  hipStream_t st = as_stream(stream);

  if (norms_dev) {
    // apply + |lhs_k|^2 in one pass: kernel A2 in fp64, nc = 1 or 2, every site written
    if (vec32 || mat32 || slab || ridx || !(nc == 1 || nc == 2) || a.par_count != 2 || lhs == rhs || nrhs > 16) return QMG_ERR_UNSUPPORTED;
    const int rows = (d->Ly % 2 == 0) ? 2 : 1;
    const unsigned gx = (unsigned)((a.hr + BLOCK / (nc * nc) - 1) / (BLOCK / (nc * nc)));
    const long ngroups = d->Ly / rows;
    const unsigned gyp = ngroups > 65535 ? 65535u : (unsigned)ngroups;
    const long nparts = (long)gyp * gx;            // one partial per block and system
    const size_t smem = sizeof(double) * BLOCK * (size_t)nrhs;
    int dev = 0;
    QMG_HIP_CHECK(hipGetDevice(&dev));
    NormWorkspace& ws = g_norm_ws;
    if (ws.device != dev || ws.cap < (size_t)nparts * nrhs) {
      if (ws.part && ws.device == dev) QMG_HIP_CHECK(hipFree(ws.part));   // (synchronises: no launch still writes the old buffer)
      ws.part = nullptr; ws.cap = 0;
      if (ws.device != dev && ws.done) { (void)hipEventDestroy(ws.done); ws.done = nullptr; }   // (an event belongs to the device it was created on)
      ws.used = false;
      QMG_HIP_CHECK(hipMalloc((void**)&ws.part, sizeof(double) * (size_t)nparts * nrhs));
      ws.cap = (size_t)nparts * nrhs; ws.device = dev;
    }
    a.norm_part = ws.part;
    if (!ws.done) QMG_HIP_CHECK(hipEventCreateWithFlags(&ws.done, hipEventDisableTiming));
    if (ws.used && ws.last != st) QMG_HIP_CHECK(hipStreamWaitEvent(st, ws.done, 0));   // the previous call's partials are still being summed on another stream
    dim3 grid(gx, gyp), block(BLOCK);
    const bool pf = nc == 1 && nrhs > 1 && g_pair_prefetch;   // (nc = 2: the prefetch costs 3 %, tools/apply_norm_ab.py)
#define QMG_NORM_LAUNCH(NC, ROWS, PF) k_stencil_pair<double, NC, ROWS, true, true, true, PF><<<grid, block, smem, st>>>(a);
    if (nc == 1) {
      if (rows == 2) { if (pf) { QMG_NORM_LAUNCH(1, 2, true) } else { QMG_NORM_LAUNCH(1, 2, false) } }
      else { if (pf) { QMG_NORM_LAUNCH(1, 1, true) } else { QMG_NORM_LAUNCH(1, 1, false) } }
    } else { if (rows == 2) { QMG_NORM_LAUNCH(2, 2, false) } else { QMG_NORM_LAUNCH(2, 1, false) } }
#undef QMG_NORM_LAUNCH
    QMG_LAUNCH_CHECK();
    k_apply_norm_final<<<nrhs, BLOCK, 0, st>>>(ws.part, nparts, norms_dev);
    QMG_LAUNCH_CHECK();
    QMG_HIP_CHECK(hipEventRecord(ws.done, st));
    ws.last = st; ws.used = true;
    return QMG_SUCCESS;
  }

  if (epi) {
    // out = other_scale other + acc_scale acc and the MR dots, in kernels B / B32 (any nc the generic kernels serve); the processed
    // parities must be overwritten (an accumulate into lhs and an `other` term at once has no single meaning)
    if (nc == 1 || nc == 2 || nc == 4) return QMG_ERR_UNSUPPORTED;   // kernels A / S / W: qmg_wilson_*_direct has its own epilogue, the rest falls back
    if ((ev && !(pieces & QMG_P_ZERO_E)) || (od && !(pieces & QMG_P_ZERO_O))) return QMG_ERR_INVALID;
    if (lhs == rhs || epi->other == lhs || epi->dotv == lhs) return QMG_ERR_INVALID;
    a.epi.on = 1;
    a.epi.other = epi->other; a.epi.other_scale = epi->other_scale; a.epi.acc_scale = epi->acc_scale;
    a.epi.dotv = epi->dotv;
    // partials: one per wavefront of the launch; the grid is fixed below (kernel B / B32: gx = ceil(hr / S), gy rows)
  }

  // fp32: the one-site-per-lane-group kernel is the faster one (4096^2 Wilson: 0.573 ms against 0.592 ms for the paired
  // kernel, profiles/r02_kernel_rooflines.json: half the bytes per site leave the paired kernel's longer dependent chain
  // exposed), so the paired kernel serves fp64 only unless "stencil_pair" asks for it explicitly (>= 8)
  const bool use_pair = vec32 ? (g_stencil_pair >= 8) : (g_stencil_pair > 0);
  if ((nc == 1 || nc == 2 || nc == 4) && a.par_count == 2 && use_pair && lhs != rhs && !slab) {
    const int E = (vec32 && nc % 2 == 0) ? nc * nc / 2 : nc * nc;   // lanes per site (KA<T, NC>::E)
    const int rows = ((g_stencil_pair & 7) >= 4 && d->Ly % 4 == 0) ? 4 : ((g_stencil_pair & 7) >= 2 && d->Ly % 2 == 0) ? 2 : 1;
    const unsigned gx = (unsigned)((a.hr + BLOCK / E - 1) / (BLOCK / E));
    unsigned gyp = (unsigned)(d->Ly / rows);
    if (gyp > 65535u) gyp = 65535u;
    if (g_stencil_rows > 0 && gyp > (unsigned)g_stencil_rows) gyp = (unsigned)g_stencil_rows;
    dim3 grid(gx, gyp), block(BLOCK);
#define QMG_PAIR_LAUNCH_T(T, NC, ROWS)                                                        \
    switch (g_stencil_nt & 3) {                                                               \
      case 0: k_stencil_pair<T, NC, ROWS, false, false><<<grid, block, 0, st>>>(a); break;    \
      case 1: k_stencil_pair<T, NC, ROWS, true, false><<<grid, block, 0, st>>>(a); break;     \
      case 2: k_stencil_pair<T, NC, ROWS, false, true><<<grid, block, 0, st>>>(a); break;     \
      default: k_stencil_pair<T, NC, ROWS, true, true><<<grid, block, 0, st>>>(a); break;     \
    }
    // staggered-type batches (nc = 1, fp64, default non-temporal policy): the variant that requests system k+1 ahead of system
    // k's arithmetic -- 4096^2, 8 systems: 1.04 -> 0.90 ms; at nc = 2 it loses 3 % (tools/apply_norm_ab.py), so not there
    const bool pf = nc == 1 && !vec32 && a.nrhs > 1 && g_pair_prefetch && (g_stencil_nt & 3) == 3;
#define QMG_PAIR_LAUNCH(NC, ROWS) if (vec32) { QMG_PAIR_LAUNCH_T(float, NC, ROWS) } else { QMG_PAIR_LAUNCH_T(double, NC, ROWS) }
#define QMG_PAIR_LAUNCH_PF(ROWS) k_stencil_pair<double, 1, ROWS, true, true, false, true><<<grid, block, 0, st>>>(a);
    if (pf) { if (rows == 4) { QMG_PAIR_LAUNCH_PF(4) } else if (rows == 2) { QMG_PAIR_LAUNCH_PF(2) } else { QMG_PAIR_LAUNCH_PF(1) } }
    else if (nc == 1) { if (rows == 4) { QMG_PAIR_LAUNCH(1, 4) } else if (rows == 2) { QMG_PAIR_LAUNCH(1, 2) } else { QMG_PAIR_LAUNCH(1, 1) } }
    if (nc == 2) { if (rows == 4) { QMG_PAIR_LAUNCH(2, 4) } else if (rows == 2) { QMG_PAIR_LAUNCH(2, 2) } else { QMG_PAIR_LAUNCH(2, 1) } }
    if (nc == 4) { if (rows >= 2) { QMG_PAIR_LAUNCH(4, 2) } else { QMG_PAIR_LAUNCH(4, 1) } }
#undef QMG_PAIR_LAUNCH
#undef QMG_PAIR_LAUNCH_PF
#undef QMG_PAIR_LAUNCH_T
    QMG_LAUNCH_CHECK();
    return QMG_SUCCESS;
  }

  if ((nc == 1 || nc == 2 || nc == 4) && !slab) {
    const int E = (vec32 && nc % 2 == 0) ? nc * nc / 2 : nc * nc;
    const unsigned gx = (unsigned)((a.hr + BLOCK / E - 1) / (BLOCK / E));
    dim3 grid(gx, gy), block(BLOCK);
#define QMG_ELEM_LAUNCH_T(T, NC)                                                              \
    switch (g_stencil_nt & 3) {                                                               \
      case 0: k_stencil_elem<T, NC, false, false><<<grid, block, 0, st>>>(a); break;          \
      case 1: k_stencil_elem<T, NC, true, false><<<grid, block, 0, st>>>(a); break;           \
      case 2: k_stencil_elem<T, NC, false, true><<<grid, block, 0, st>>>(a); break;           \
      default: k_stencil_elem<T, NC, true, true><<<grid, block, 0, st>>>(a); break;           \
    }
#define QMG_ELEM_LAUNCH(NC) if (vec32) { QMG_ELEM_LAUNCH_T(float, NC) } else { QMG_ELEM_LAUNCH_T(double, NC) }
    if (nc == 1) { QMG_ELEM_LAUNCH(1) }
    if (nc == 2) { QMG_ELEM_LAUNCH(2) }
    if (nc == 4) { QMG_ELEM_LAUNCH(4) }
#undef QMG_ELEM_LAUNCH
#undef QMG_ELEM_LAUNCH_T
    QMG_LAUNCH_CHECK();
    return QMG_SUCCESS;
  }

  // several right-hand sides against one matrix read: kernel C (f64 MFMA) from 4 systems up -- measured 512^2 nc = 24, 8 rhs:
  // 2.84 ms against 4.84 ms for the vector-FMA kernel B, which tops out near 10 TFLOP/s on LDS traffic; with 2-3 systems
  // kernel B's shared tile wins (nc = 8, 1024^2, 3 rhs: 1.06 vs 1.39 ms) and it serves every other nc
  // (nc <= 16: kernel B with one 4-accumulator pass still wins at exactly 4 systems -- nc = 8, 1024^2: 1.21 vs 1.52 ms;
  //  nc = 16, 512^2: 0.98 vs 1.08 ms -- so there the matrix cores take over from 5)
  if (a.nrhs >= (nc <= 16 ? 5 : 4) && g_stencil_mfma && (nc == 8 || nc == 12 || nc == 16 || nc == 24 || nc == 32)) {
    // kernel C: up to 16 right-hand sides per pass share one read of the matrices
    const unsigned gx = (unsigned)((a.hr + BLOCK / WAVE - 1) / (BLOCK / WAVE));
    dim3 grid(gx, gy), block(BLOCK);
    for (int k0 = 0; k0 < a.nrhs; k0 += 16) {
      StencilArgs b = a;
      b.lhs = (char*)a.lhs + (size_t)k0 * a.vec_stride * (vec32 ? 8 : 16);
      b.rhs = (const char*)a.rhs + (size_t)k0 * a.vec_stride * (vec32 ? 8 : 16);
      const int nk = (a.nrhs - k0 < 16) ? a.nrhs - k0 : 16;
      size_t smem = a.mat32 ? sizeof(float2) * (size_t)(BLOCK / WAVE) * nc * (nc + 2) : sizeof(cplx) * (size_t)(BLOCK / WAVE) * nc * (nc + 1);
      int mode = (g_stencil_mfma == 2 || nk > 8) ? 0 : 1;
      // 9-16 systems in fp64: the real-form tiles where they save MFMAs (nc = 24: 36 instead of 48 per piece; nc = 8: 4 instead of 8)
      if (mode == 0 && g_stencil_mfma == 1 && g_mfma_vl && !a.mat32 && !a.vec32 && (nc == 24 || nc == 8)) mode = 2;
      const bool vl_slices = a.mat16 ? (mode == 1) : (g_mfma_vl && !(mode == 0 && a.mat32));   // (the 16-bit instantiations: VL with MODE 1, not with MODE 0)
      if (vl_slices) smem += sizeof(cplx) * (size_t)(BLOCK / WAVE) * (mode == 1 ? 8 : 16) * (nc + 1);   // the wavefronts' vector slices
#define QMG_MFMA_LAUNCH0(NC, MODE, M32, V32, VL)                                                              \
      {                                                                                                         \
        if (smem > 64 * 1024)                                                                                   \
          QMG_HIP_CHECK(hipFuncSetAttribute((const void*)k_stencil_mfma<NC, MODE, M32, V32, VL>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem)); \
        k_stencil_mfma<NC, MODE, M32, V32, VL><<<grid, block, smem, st>>>(b, nk);                             \
      }
#define QMG_MFMA_LAUNCH1(NC, MODE, M32, V32) { if (g_mfma_vl && !(MODE == 0 && M32)) QMG_MFMA_LAUNCH0(NC, MODE, M32, V32, true) else QMG_MFMA_LAUNCH0(NC, MODE, M32, V32, false) }
#define QMG_MFMA_LAUNCH16(NC, MODE, V32)                                                                        \
      {                                                                                                         \
        if (smem > 64 * 1024)                                                                                   \
          QMG_HIP_CHECK(hipFuncSetAttribute((const void*)k_stencil_mfma<NC, MODE, true, V32, MODE == 1, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem)); \
        k_stencil_mfma<NC, MODE, true, V32, MODE == 1, true><<<grid, block, smem, st>>>(b, nk);               \
      }
#define QMG_MFMA_LAUNCH2(NC, MODE)                                                                              \
      { if (a.mat16) { if (a.vec32) QMG_MFMA_LAUNCH16(NC, MODE, true) else QMG_MFMA_LAUNCH16(NC, MODE, false) }  \
        else if (a.vec32) QMG_MFMA_LAUNCH1(NC, MODE, true, true) else if (a.mat32) QMG_MFMA_LAUNCH1(NC, MODE, true, false) else QMG_MFMA_LAUNCH1(NC, MODE, false, false) }
#define QMG_MFMA_LAUNCH(NC)                                                                                     \
      if (mode == 2) QMG_MFMA_LAUNCH0(NC, 2, false, false, true) else if (mode == 0) QMG_MFMA_LAUNCH2(NC, 0) else QMG_MFMA_LAUNCH2(NC, 1)
      // nc = 8, up to 8 systems, whole lattice: two sites per wavefront (PAIR; "mfma_pair8" = 0 keeps one)
      if (nc == 8 && mode == 1 && g_mfma_pair8 && !slab && (a.hr % 2 == 0)) {
        const unsigned gxp = (unsigned)((a.hr / 2 + BLOCK / WAVE - 1) / (BLOCK / WAVE));
        dim3 gridp(gxp, gy);
        const size_t smemp = (a.mat32 ? sizeof(float2) * (size_t)(BLOCK / WAVE) * 16 * 18 : sizeof(cplx) * (size_t)(BLOCK / WAVE) * 16 * 17) +
                             sizeof(cplx) * (size_t)(BLOCK / WAVE) * 8 * 17;
        if (a.mat16) { if (a.vec32) k_stencil_mfma<16, 1, true, true, true, true, true><<<gridp, block, smemp, st>>>(b, nk);
                       else k_stencil_mfma<16, 1, true, false, true, true, true><<<gridp, block, smemp, st>>>(b, nk); }
        else if (a.vec32) k_stencil_mfma<16, 1, true, true, true, false, true><<<gridp, block, smemp, st>>>(b, nk);
        else if (a.mat32) k_stencil_mfma<16, 1, true, false, true, false, true><<<gridp, block, smemp, st>>>(b, nk);
        else k_stencil_mfma<16, 1, false, false, true, false, true><<<gridp, block, smemp, st>>>(b, nk);
        continue;
      }
      switch (nc) {
        case 8: QMG_MFMA_LAUNCH(8) break;
        case 12: QMG_MFMA_LAUNCH(12) break;
        case 16: QMG_MFMA_LAUNCH(16) break;
        case 24: QMG_MFMA_LAUNCH(24) break;
        default: QMG_MFMA_LAUNCH(32) break;
      }
#undef QMG_MFMA_LAUNCH
#undef QMG_MFMA_LAUNCH1
#undef QMG_MFMA_LAUNCH0
#undef QMG_MFMA_LAUNCH2
#undef QMG_MFMA_LAUNCH16
    }
    QMG_LAUNCH_CHECK();
    return QMG_SUCCESS;
  }

  if (nc > BLOCK) return QMG_ERR_UNSUPPORTED;
  if (a.mat32 && !(nc & 1) && g_gen32 && !(slab && nc <= 4)) {   // (a slab's fp32 applies at nc = 4 keep kernel B's widening loads)
    // kernel B32: fp32 tile end to end (even nc)
    const GenLayout L = make_gen_layout(nc, a.hr, g_gen32 == 2 ? 1 : 0);
    if (a.mat16 && (nc & 3)) return QMG_ERR_UNSUPPORTED;
    const int pp = a.mat16 ? (L.mat_elems / 4 + BLOCK - 1) / BLOCK : (L.mat_elems / 2 + BLOCK - 1) / BLOCK;
    if (pp >= 1 && pp <= (a.mat16 ? 3 : 6)) {
      int kr = (a.nrhs >= 5) ? 8 : (a.nrhs >= 2) ? 4 : 1;
      auto smem_of = [&](int k) { return (((size_t)L.S * nc * (nc + 2) * 8 + 15) & ~(size_t)15) + sizeof(cplx) * ((size_t)k * L.S * nc + (size_t)L.H * L.S * nc); };
      while (kr > 1 && smem_of(kr) > 48 * 1024) kr = (kr == 8) ? 4 : 1;
      const size_t smem = smem_of(kr);
      if (smem <= 64 * 1024) {
        const unsigned gx = (unsigned)((a.hr + L.S - 1) / L.S);
        dim3 grid(gx, gy), block(BLOCK);
        QMG_EPI_BEGIN(gx, gy)
#define QMG_G32_CASE2(PP, KR) { if (a.vec32) k_stencil_gen32<PP, KR, true><<<grid, block, smem, st>>>(a, nc, L); else k_stencil_gen32<PP, KR, false><<<grid, block, smem, st>>>(a, nc, L); }
#define QMG_G32_EPI(PP) { if (a.vec32) k_stencil_gen32<PP, 1, true, true><<<grid, block, smem, st>>>(a, nc, L); else k_stencil_gen32<PP, 1, false, true><<<grid, block, smem, st>>>(a, nc, L); }
#define QMG_G32_CASE(PP) case PP: { if (kr == 8) { QMG_G32_CASE2(PP, 8) } else if (kr == 4) { QMG_G32_CASE2(PP, 4) } else if (a.epi.on) { QMG_G32_EPI(PP) } else { QMG_G32_CASE2(PP, 1) } } break;
#define QMG_G16_CASE2(PP, KR) { if (a.vec32) k_stencil_gen32<PP, KR, true, false, true><<<grid, block, smem, st>>>(a, nc, L); else k_stencil_gen32<PP, KR, false, false, true><<<grid, block, smem, st>>>(a, nc, L); }
#define QMG_G16_EPI(PP) { if (a.vec32) k_stencil_gen32<PP, 1, true, true, true><<<grid, block, smem, st>>>(a, nc, L); else k_stencil_gen32<PP, 1, false, true, true><<<grid, block, smem, st>>>(a, nc, L); }
#define QMG_G16_CASE(PP) case PP: { if (kr == 8) { QMG_G16_CASE2(PP, 8) } else if (kr == 4) { QMG_G16_CASE2(PP, 4) } else if (a.epi.on) { QMG_G16_EPI(PP) } else { QMG_G16_CASE2(PP, 1) } } break;
        if (a.mat16) { switch (pp) { QMG_G16_CASE(1) QMG_G16_CASE(2) QMG_G16_CASE(3) default: break; } }
        else
        switch (pp) { QMG_G32_CASE(1) QMG_G32_CASE(2) QMG_G32_CASE(3) QMG_G32_CASE(4) QMG_G32_CASE(5) QMG_G32_CASE(6) default: break; }
#undef QMG_G16_CASE
#undef QMG_G16_EPI
#undef QMG_G16_CASE2
#undef QMG_G32_CASE
#undef QMG_G32_EPI
#undef QMG_G32_CASE2
        QMG_LAUNCH_CHECK();
        QMG_EPI_FINISH()
        return QMG_SUCCESS;
      }
    }
  }
  if (a.mat16) return QMG_ERR_UNSUPPORTED;   // complex<half> matrices are served by kernels B32 / C only (nc a multiple of 4)
  GenLayout L = make_gen_layout(nc, a.hr, a.mat32);
  if (L.per_thread > GEN_MAX_PER_THREAD) return QMG_ERR_UNSUPPORTED;   // nc > 55: S = 1 still too large
  // right-hand sides per pass of kernel B: 4 (2-4 systems) or 8 accumulators; if the tile plus the vectors of the pass do
  // not fit 64 KB of LDS (>= 2 blocks per CU) the tile shrinks first (nc = 16: 12 -> 6 sites), the pass second
  int kr = (a.nrhs >= 5) ? 8 : (a.nrhs >= 2) ? 4 : 1;
  auto smem_of = [&](const GenLayout& l, int k) { return sizeof(cplx) * ((size_t)l.S * nc * l.rs + (size_t)k * l.S * nc + (size_t)l.H * l.S * nc); };
  while (kr > 1 && smem_of(L, kr) > 64 * 1024 && L.S > 1) L = make_gen_layout(nc, a.hr, a.mat32, (L.S + 1) / 2);
  while (kr > 1 && smem_of(L, kr) > 64 * 1024) kr = (kr == 8) ? 4 : 1;
  const size_t smem = smem_of(L, kr);
  if (smem > 160 * 1024) return QMG_ERR_UNSUPPORTED;
  const unsigned gx = (unsigned)((a.hr + L.S - 1) / L.S);
  dim3 grid(gx, gy), block(BLOCK);
  QMG_EPI_BEGIN(gx, gy)
#define QMG_GEN_CASE3(PT, M32, KR, V32)                                                                 \
    {                                                                                                   \
      if (smem > 64 * 1024)                                                                             \
        QMG_HIP_CHECK(hipFuncSetAttribute((const void*)k_stencil_gen<PT, M32, KR, V32>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem)); \
      k_stencil_gen<PT, M32, KR, V32><<<grid, block, smem, st>>>(a, nc, L);                             \
    }
#define QMG_GEN_CASE2(PT, M32, KR) { if (M32 && a.vec32) QMG_GEN_CASE3(PT, true, KR, true) else QMG_GEN_CASE3(PT, M32, KR, false) }
#define QMG_GEN_EPI3(PT, M32, V32)                                                                      \
    {                                                                                                   \
      if (smem > 64 * 1024)                                                                             \
        QMG_HIP_CHECK(hipFuncSetAttribute((const void*)k_stencil_gen<PT, M32, 1, V32, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem)); \
      k_stencil_gen<PT, M32, 1, V32, true><<<grid, block, smem, st>>>(a, nc, L);                        \
    }
#define QMG_GEN_EPI(PT, M32) { if (M32 && a.vec32) QMG_GEN_EPI3(PT, true, true) else QMG_GEN_EPI3(PT, M32, false) }
#define QMG_GEN_CASE1(PT, M32)                                                                          \
    { if (kr == 8) QMG_GEN_CASE2(PT, M32, 8) else if (kr == 4) QMG_GEN_CASE2(PT, M32, 4) else if (a.epi.on) QMG_GEN_EPI(PT, M32) else QMG_GEN_CASE2(PT, M32, 1) }
#define QMG_GEN_CASE(PT)                                                                                \
  case PT:                                                                                              \
    if (a.mat32) QMG_GEN_CASE1(PT, true) else QMG_GEN_CASE1(PT, false)                                  \
    break;
  switch (L.per_thread) {
    QMG_GEN_CASE(1) QMG_GEN_CASE(2) QMG_GEN_CASE(3) QMG_GEN_CASE(4) QMG_GEN_CASE(5) QMG_GEN_CASE(6)
    QMG_GEN_CASE(7) QMG_GEN_CASE(8) QMG_GEN_CASE(9) QMG_GEN_CASE(10) QMG_GEN_CASE(11) QMG_GEN_CASE(12)
    default: return QMG_ERR_UNSUPPORTED;
  }
#undef QMG_GEN_CASE
#undef QMG_GEN_CASE1
#undef QMG_GEN_CASE2
#undef QMG_GEN_EPI
#undef QMG_GEN_EPI3
#undef QMG_GEN_CASE3
  QMG_LAUNCH_CHECK();
  QMG_EPI_FINISH()
  return QMG_SUCCESS;
}

// Generic-nc slab apply (csrc/qmg_site.hip holds the C entry qmg_stencil_apply_slab and serves nc = 2 itself): kernels B / B32 / C with the
// right-hand side's rows -1 / Ly from the halo buffers.  mat32: 0 fp64 matrices, 1 complex<float>, 2 complex<half>; vec32: complex<float> vectors.
int qmg::generic_slab_apply(const qmg_stencil_desc* d, void* lhs, const void* rhs, unsigned pieces, int n, long vec_stride, const unsigned char* ridx,
                            hipStream_t st, const SlabHalo* slab, int mat32, int vec32) {
  return stencil_apply_impl(d, lhs, rhs, pieces, n, (size_t)vec_stride, ridx, (void*)st, mat32, vec32, slab);
}

// qmg_site.hip -- the fine Wilson-type apply (nc = 2) as a "site kernel": every global access of a lane is ONE 16-byte chunk,
// in three storage precisions (kernel S of DESIGN.md):
//   ST 2: complex<double> matrices and vectors, fp64 arithmetic      384 B/site   4 lanes per site (lane = one matrix element)
//   ST 1: complex<float>  matrices and vectors, fp32 arithmetic      192 B/site   2 lanes per site (lane = one matrix row)
//   ST 0: complex<half>   matrices, complex<float> vectors, fp32     112 B/site   1 lane per site  (lane = the 2x2 matrix)
// ST 0 is the "16-bit-storage smoother" of SURVEY 8f-4 (not in the reference, which is fp64 only).
//
// Why 16-bit matrices only: 5/6 of the fp32 apply's 192 B/site are matrices (160 B); in half precision they are 80 B, so
// the apply moves 80 + 16 + 16 = 112 B/site while the vectors, whose magnitude falls by ten orders of magnitude over a
// solve, keep fp32's range.  The entries of a Wilson-type operator are O(1) (2w on the clover diagonal, +-1/2 U in the hops,
// C^-1 H after right-block-Jacobi), well inside fp16's range; the rounding (2^-11) perturbs the OPERATOR by ~5e-4, harmless
// inside the K-cycle that preconditions a flexible fp64 outer solve.
//
// What makes the kernel fast (measured on ST 0 first: 0.55 ms -> 0.27 ms at 4096^2, 3.4 -> 7.0 TB/s):
//   * the piece set is uniform per launch, but as RUN-TIME flags it puts every load behind a scalar branch and the compiler
//     drains the memory queue (s_waitcnt vmcnt(0)) at each join.  The shapes the K-cycle uses are therefore compile-time:
//       SHAPE 1: clover + four hops of each processed parity (M, and the parity-restricted M of the Schur steps)
//       SHAPE 2: four hops, no clover (D_eo / D_oe, the right-block-Jacobi hops)
//       SHAPE 0: anything else (run-time flags);
//   * a load phase of ten independent 16-byte loads per lane kept as RAW registers (widening / products afterwards), fenced
//     by a scheduling barrier -- without it the machine scheduler sinks loads into the arithmetic to shorten live ranges;
//   * the systems of a batch are looped INSIDE (matrices stay in registers: 320 + 64 n bytes per site, not 384 n; the other
//     resident wavefronts cover a system's five-load phase); their slot numbers are ints in the kernel arguments (a byte
//     table turns a uniform index into global_load_ubyte + s_waitcnt vmcnt(0) in front of every system).
// Rows of both parities are interleaved in block order (row & 1 = parity), so the second use of a right-hand-side row hits L2.
#include <hip/hip_fp16.h>

#include "qmg_common.h"

namespace qmg {

struct SiteArgs {
  const void* clover;      // 2x2 matrices in the storage type, or NULL
  const void* hopping;
  void* lhs;
  const void* rhs;
  int hr, Ly;
  long half_vol, size_cm;  // size_cm in matrix ELEMENTS (4 per site and direction)
  unsigned pieces;
  int nrhs;
  long vec_stride;         // complex elements between right-hand sides
  int par_first, par_count, nrows;
  double shift[2], eo_shift[2], dof_shift[2];
  int ridx[16];
  // y-slab of a larger lattice (qmg_stencil_apply_slab): rows -1 and Ly of the right-hand side come from these buffers
  // ([system][parity][hr] site vectors) instead of the periodic wrap; NULL = periodic in y
  const void* halo_lo;
  const void* halo_hi;
  long halo_stride;        // complex elements between the systems of a halo buffer
  int y_first, y_count;    // the rows of this launch: y_first .. y_first + y_count - 1 ...
  int boundary_only;       // ... or, if set, the two boundary rows y = 0 and y = Ly - 1 (y_count = 2)
};

typedef float v4f __attribute__((ext_vector_type(4)));
typedef double v2d __attribute__((ext_vector_type(2)));

template <int ST> struct SiteT;
template <> struct SiteT<0> { typedef float R; static constexpr int LPS = 1; };
template <> struct SiteT<1> { typedef float R; static constexpr int LPS = 2; };
template <> struct SiteT<2> { typedef double R; static constexpr int LPS = 4; };

__device__ __forceinline__ v4f ld16(const void* p, long chunk) { return *(reinterpret_cast<const v4f*>(p) + chunk); }
__device__ __forceinline__ v4f ld16_nt(const void* p, long chunk) { return __builtin_nontemporal_load(reinterpret_cast<const v4f*>(p) + chunk); }

template <typename R>
__device__ __forceinline__ void fmac2(R& ax, R& ay, R mx, R my, R bx, R by) {
  ax = fma(mx, bx, ax); ax = fma(-my, by, ax);
  ay = fma(mx, by, ay); ay = fma(my, bx, ay);
}
__device__ __forceinline__ v2d as_d(const v4f raw) { return __builtin_bit_cast(v2d, raw); }

// One term: acc += (this lane's part of the 2x2 matrix) * (the right-hand-side chunk).  Accumulators:
//   ST 0: acc[0..3] = out[0].re, out[0].im, out[1].re, out[1].im       ST 1: acc[0..1] = out[row]
//   ST 2: acc[0..1] = partial of out[row] from column c (summed over the lane pair at the end)
template <int ST, typename R>
__device__ __forceinline__ void term(R* acc, const v4f m, const v4f x) {
  if constexpr (ST == 0) {
    const __half2* h = reinterpret_cast<const __half2*>(&m);
    const float2 m00 = __half22float2(h[0]), m01 = __half22float2(h[1]), m10 = __half22float2(h[2]), m11 = __half22float2(h[3]);
    fmac2<float>(acc[0], acc[1], m00.x, m00.y, x.x, x.y); fmac2<float>(acc[0], acc[1], m01.x, m01.y, x.z, x.w);
    fmac2<float>(acc[2], acc[3], m10.x, m10.y, x.x, x.y); fmac2<float>(acc[2], acc[3], m11.x, m11.y, x.z, x.w);
  } else if constexpr (ST == 1) {
    fmac2<float>(acc[0], acc[1], m.x, m.y, x.x, x.y); fmac2<float>(acc[0], acc[1], m.z, m.w, x.z, x.w);
  } else {
    const v2d md = as_d(m), xd = as_d(x);
    fmac2<double>(acc[0], acc[1], md.x, md.y, xd.x, xd.y);
  }
}

// The right-hand-side chunks of one system for one site group: four neighbours, the site itself, and what the lhs holds.
template <int NACC, typename R>
struct SiteX { v4f nb[4], own; R prev[NACC]; };

template <int ST, int SHAPE, bool ZERO, bool BATCH>
__global__ __launch_bounds__(BLOCK) void k_stencil_site(const SiteArgs a) {
  typedef typename SiteT<ST>::R R;
  constexpr int LPS = SiteT<ST>::LPS;
  constexpr int NACC = (ST == 0) ? 4 : 2;
  constexpr int XCH = (ST == 2) ? 2 : 1;          // 16-byte chunks per site of a vector
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  const int j = t / LPS, sub = t % LPS;
  if (j >= a.hr) return;
  const int xc = (ST == 2) ? (sub & 1) : 0;       // which vector chunk of a site this lane multiplies with
  const int orow = (ST == 2) ? (sub >> 1) : sub;  // which output row it contributes to (ST 0: both)
  const long sys_bytes = a.vec_stride * (long)(2 * sizeof(R));
  for (int row = blockIdx.y; row < a.nrows; row += gridDim.y) {
    const int p = (a.par_count == 2) ? (row & 1) : a.par_first;
    const int yi = (a.par_count == 2) ? (row >> 1) : row;
    const int y = a.boundary_only ? (yi ? a.Ly - 1 : 0) : a.y_first + yi;
    const bool do_clover = SHAPE == 1 || (SHAPE == 0 && a.clover && ((a.pieces >> p) & 1u));
    const unsigned hop_mask = SHAPE ? 0xFu : (a.hopping ? ((a.pieces >> (2 + 4 * p)) & 0xFu) : 0u);
    const bool do_shift = (a.pieces >> (10 + p)) & 1u;
    const bool do_zero = ZERO || ((a.pieces >> (12 + p)) & 1u);
    const bool need_own = SHAPE == 1 || do_clover || do_shift;
    const long site = (long)p * a.half_vol + (long)y * a.hr + j;
    const long opp = (long)(1 - p) * a.half_vol;
    const int s = (y + p) & 1;
    int jp = j + s;     if (jp == a.hr) jp = 0;
    int jm = j + s - 1; if (jm < 0) jm = a.hr - 1;
    const int yp = (y + 1 == a.Ly) ? 0 : y + 1;
    const int ym = (y == 0) ? a.Ly - 1 : y - 1;
    const long nb[4] = {opp + (long)y * a.hr + jp, opp + (long)yp * a.hr + j, opp + (long)y * a.hr + jm, opp + (long)ym * a.hr + j};
    // what this lane stores: ST 0 the site's 16 bytes; ST 1 its row (8 bytes); ST 2 one double (component xc of row orow)
    const long out_elem = (ST == 0) ? site : (ST == 1) ? site * 2 + sub : (site * 2 + orow) * 2 + xc;

    // one system's right-hand-side chunks, as raw registers
    auto load_x = [&](SiteX<NACC, R>& v, int k) {
      const long off = (long)a.ridx[k] * sys_bytes;
      const char* x = reinterpret_cast<const char*>(a.rhs) + off;
      // a slab's rows -1 / Ly: the opposite-parity row of the halo buffer (row-uniform, so a scalar select)
      const bool from_hi = a.halo_hi && y + 1 == a.Ly, from_lo = a.halo_lo && y == 0;
      const long hoff = (long)a.ridx[k] * a.halo_stride * (long)(2 * sizeof(R));
      const long hsite = (long)(1 - p) * a.hr + j;
#pragma unroll
      for (int d = 0; d < 4; d++) {
        if (!((hop_mask >> d) & 1u)) continue;
        if (d == 1 && from_hi) v.nb[d] = ld16(reinterpret_cast<const char*>(a.halo_hi) + hoff, hsite * XCH + xc);
        else if (d == 3 && from_lo) v.nb[d] = ld16(reinterpret_cast<const char*>(a.halo_lo) + hoff, hsite * XCH + xc);
        else v.nb[d] = ld16(x, nb[d] * XCH + xc);
      }
      if (need_own) v.own = ld16(x, site * XCH + xc);
      if (!do_zero) {
        const char* o = reinterpret_cast<const char*>(a.lhs) + off;
        if constexpr (ST == 0) { const v4f q = ld16(o, out_elem); v.prev[0] = q.x; v.prev[1] = q.y; v.prev[2] = q.z; v.prev[3] = q.w; }
        else if constexpr (ST == 1) { const float2 q = *(reinterpret_cast<const float2*>(o) + out_elem); v.prev[0] = q.x; v.prev[1] = q.y; }
        else v.prev[0] = *(reinterpret_cast<const R*>(o) + out_elem);
      }
    };

    // ---- load phase: the first system's chunks, then the matrices (once per site, for every system of a batch).
    // The system comes first so that nothing the compiler hoists out of the batch loop (the fp16 widening) can sit
    // between the two groups of loads.  BATCH = false is the single-system code without the loop (fewer registers:
    // 0.27 ms against 0.31 ms for the 16-bit apply at 4096^2).
    const int nsys = BATCH ? a.nrhs : 1;
    SiteX<NACC, R> cur;
    load_x(cur, 0);
    v4f mc, mh[4];
    if (do_clover) mc = ld16_nt(a.clover, site * LPS + sub);
#pragma unroll
    for (int d = 0; d < 4; d++)
      if ((hop_mask >> d) & 1u) mh[d] = ld16_nt(a.hopping, ((long)d * (a.size_cm / 4) + site) * LPS + sub);
    for (int k = 0; k < nsys; k++) {
      if (BATCH && k > 0) load_x(cur, k);
      __builtin_amdgcn_sched_barrier(0);            // keep the scheduler from sinking loads into the arithmetic
      // ---- products (clover first, then +x, +y, -x, -y, as the reference's loop order) and the store
      R acc[NACC];
#pragma unroll
      for (int q = 0; q < NACC; q++) acc[q] = R(0);
      if (do_clover) term<ST, R>(acc, mc, cur.own);
#pragma unroll
      for (int d = 0; d < 4; d++)
        if ((hop_mask >> d) & 1u) term<ST, R>(acc, mh[d], cur.nb[d]);
      if (do_shift) {   // shift +- eo_shift +- dof_shift on the diagonal (stencil_2d.h:890-908); nc = 2: dof sign +, -
        const double sg = p ? -1.0 : 1.0;
        const R s0x = (R)(a.shift[0] + sg * a.eo_shift[0] + a.dof_shift[0]), s0y = (R)(a.shift[1] + sg * a.eo_shift[1] + a.dof_shift[1]);
        const R s1x = (R)(a.shift[0] + sg * a.eo_shift[0] - a.dof_shift[0]), s1y = (R)(a.shift[1] + sg * a.eo_shift[1] - a.dof_shift[1]);
        const v4f own = cur.own;
        if constexpr (ST == 0) {
          fmac2<R>(acc[0], acc[1], s0x, s0y, (R)own.x, (R)own.y);
          fmac2<R>(acc[2], acc[3], s1x, s1y, (R)own.z, (R)own.w);
        } else if constexpr (ST == 1) {
          if (sub == 0) fmac2<R>(acc[0], acc[1], s0x, s0y, (R)own.x, (R)own.y);
          else fmac2<R>(acc[0], acc[1], s1x, s1y, (R)own.z, (R)own.w);
        } else {
          const v2d od = as_d(own);                  // x[xc]: the diagonal term lives on the lane with column == row
          if (sub == 0) fmac2<R>(acc[0], acc[1], s0x, s0y, (R)od.x, (R)od.y);
          if (sub == 3) fmac2<R>(acc[0], acc[1], s1x, s1y, (R)od.x, (R)od.y);
        }
      }
      char* out = reinterpret_cast<char*>(a.lhs) + (long)a.ridx[k] * sys_bytes;
      if constexpr (ST == 0) {
        if (!do_zero) { acc[0] += cur.prev[0]; acc[1] += cur.prev[1]; acc[2] += cur.prev[2]; acc[3] += cur.prev[3]; }
        v4f o;
        o.x = (float)acc[0]; o.y = (float)acc[1]; o.z = (float)acc[2]; o.w = (float)acc[3];
        __builtin_nontemporal_store(o, reinterpret_cast<v4f*>(out) + out_elem);
      } else if constexpr (ST == 1) {
        if (!do_zero) { acc[0] += cur.prev[0]; acc[1] += cur.prev[1]; }
        typedef float v2f __attribute__((ext_vector_type(2)));
        v2f o;
        o.x = (float)acc[0]; o.y = (float)acc[1];
        __builtin_nontemporal_store(o, reinterpret_cast<v2f*>(out) + out_elem);
      } else {
        // the two column partials of a row sit in neighbouring lanes: one DPP exchange, then lane xc stores component xc,
        // so the wavefront's store is 64 consecutive doubles
        const double tx = (double)acc[0] + lane_xor1((double)acc[0]);
        const double ty = (double)acc[1] + lane_xor1((double)acc[1]);
        double o = xc ? ty : tx;
        if (!do_zero) o += (double)cur.prev[0];
        __builtin_nontemporal_store(o, reinterpret_cast<double*>(out) + out_elem);
      }
    }
  }
}

template <typename TS>
__global__ __launch_bounds__(BLOCK) void k_to_half(__half2* __restrict__ dst, const void* __restrict__ src, long n) {
  for (long i = (long)blockIdx.x * BLOCK + threadIdx.x; i < n; i += (long)gridDim.x * BLOCK) {
    const cplx v = ldc<TS>(src, i);
    dst[i] = __floats2half2_rn((float)v.x, (float)v.y);
  }
}

template <typename TD>
__global__ __launch_bounds__(BLOCK) void k_from_half(void* __restrict__ dst, const __half2* __restrict__ src, long n) {
  for (long i = (long)blockIdx.x * BLOCK + threadIdx.x; i < n; i += (long)gridDim.x * BLOCK) {
    const float2 f = __half22float2(src[i]);
    stc<TD>(dst, i, cmake((double)f.x, (double)f.y));
  }
}

int g_site_block = BLOCK, g_site_gy = 0;
int g_site_generic = 0;   // "site_generic": 1 forces the run-time-flag kernel (SHAPE 0)

template <int ST, bool BATCH>
static void launch_site_b(const SiteArgs& a, int shape, bool zero, dim3 grid, hipStream_t st) {
  const int BLOCK = g_site_block;
  if (shape == 1) { if (zero) k_stencil_site<ST, 1, true, BATCH><<<grid, BLOCK, 0, st>>>(a); else k_stencil_site<ST, 1, false, BATCH><<<grid, BLOCK, 0, st>>>(a); }
  else if (shape == 2) { if (zero) k_stencil_site<ST, 2, true, BATCH><<<grid, BLOCK, 0, st>>>(a); else k_stencil_site<ST, 2, false, BATCH><<<grid, BLOCK, 0, st>>>(a); }
  else k_stencil_site<ST, 0, false, BATCH><<<grid, BLOCK, 0, st>>>(a);
}
template <int ST>
static void launch_site(const SiteArgs& a, int shape, bool zero, dim3 grid, hipStream_t st) {
  if (a.nrhs == 1) launch_site_b<ST, false>(a, shape, zero, grid, st);
  else launch_site_b<ST, true>(a, shape, zero, grid, st);
}

// storage: 0 = complex<half> matrices + complex<float> vectors, 1 = complex<float>, 2 = complex<double>.  `ridx` = the slots
// of the n right-hand sides (NULL: 0..n-1).  nc = 2 only.  `only_where_faster`: return SITE_DECLINED for the launches that
// kernel A of qmg_stencil.hip does as well or better -- measured at 4096^2 (tools/h16_shapes.py): fp64 M 1.10 ms both,
// fp64 batches of 8 0.63 ms (A) against 0.78 ms, fp64 D_eo 0.395 ms (site) against 0.425 ms.
int site_kernel_apply(int storage, const qmg_stencil_desc* d, void* lhs, const void* rhs, unsigned pieces, int n, long vec_stride,
                      const unsigned char* ridx, hipStream_t st, bool only_where_faster, const SlabHalo* slab) {
  if (d->nc != 2 || n < 1 || n > 16 || storage < 0 || storage > 2) return QMG_ERR_UNSUPPORTED;
  SiteArgs a;
  a.clover = d->clover; a.hopping = d->hopping; a.lhs = lhs; a.rhs = rhs;
  a.hr = d->Lx / 2; a.Ly = d->Ly;
  a.half_vol = (long)a.hr * d->Ly;
  a.size_cm = 2 * a.half_vol * 4;
  a.pieces = pieces;
  a.vec_stride = vec_stride;
  a.nrhs = n;
  for (int k = 0; k < 16; k++) a.ridx[k] = (k < n) ? (ridx ? (int)ridx[k] : k) : 0;
  for (int i = 0; i < 2; i++) { a.shift[i] = d->shift[i]; a.eo_shift[i] = d->eo_shift[i]; a.dof_shift[i] = d->dof_shift[i]; }
  const unsigned even_bits = QMG_P_CLOVER_E | QMG_P_EO | QMG_P_SHIFT_E | QMG_P_ZERO_E;
  const unsigned odd_bits = QMG_P_CLOVER_O | QMG_P_OE | QMG_P_SHIFT_O | QMG_P_ZERO_O;
  const bool ev = pieces & even_bits, od = pieces & odd_bits;
  if (!ev && !od) return QMG_SUCCESS;
  a.par_first = ev ? 0 : 1;
  a.par_count = (ev && od) ? 2 : 1;
  a.halo_lo = slab ? slab->lo : nullptr;
  a.halo_hi = slab ? slab->hi : nullptr;
  a.halo_stride = slab ? slab->stride : 0;
  a.boundary_only = slab && slab->rows == 2;
  a.y_first = (slab && slab->rows == 1) ? 1 : 0;
  a.y_count = a.boundary_only ? 2 : (slab && slab->rows == 1) ? d->Ly - 2 : d->Ly;
  if (a.y_count <= 0) return QMG_SUCCESS;
  a.nrows = a.y_count * a.par_count;
  // the compile-time shape, if every processed parity asks for the same complete set
  int sh[2] = {0, 0};
  bool zero = true;
  for (int q = 0; q < a.par_count; q++) {
    const int p = (a.par_count == 2) ? q : a.par_first;
    const bool cl = d->clover && ((pieces >> p) & 1u);
    const unsigned hm = d->hopping ? ((pieces >> (2 + 4 * p)) & 0xFu) : 0u;
    sh[q] = (hm == 0xFu) ? (cl ? 1 : 2) : 0;
    if (!((pieces >> (12 + p)) & 1u)) zero = false;
  }
  int shape = (a.par_count == 2 && sh[0] != sh[1]) ? 0 : sh[0];
  if (only_where_faster && storage == 2 && !(n == 1 && shape == 2)) return SITE_DECLINED;
  if (g_site_generic) shape = 0;
  const int lps = storage == 0 ? 1 : storage == 1 ? 2 : 4;
  const long lanes = (long)a.hr * lps;
  unsigned gy = a.nrows > 65535 ? 65535u : (unsigned)a.nrows;
  if (g_site_gy > 0 && gy > (unsigned)g_site_gy) gy = (unsigned)g_site_gy;
  dim3 grid((unsigned)((lanes + g_site_block - 1) / g_site_block), gy);
  if (storage == 0) launch_site<0>(a, shape, zero, grid, st);
  else if (storage == 1) launch_site<1>(a, shape, zero, grid, st);
  else launch_site<2>(a, shape, zero, grid, st);
  QMG_LAUNCH_CHECK();
  return QMG_SUCCESS;
}

}  // namespace qmg

using namespace qmg;

extern "C" {

// complex<double> or complex<float> (src_dtype) -> complex<half>, round to nearest
int qmg_convert_to_c16(void* dst_c16, const void* src, int src_dtype, size_t n, void* stream) {
  if (!valid_dtype(src_dtype) || ((!dst_c16 || !src) && n)) return QMG_ERR_INVALID;
  if (n == 0) return QMG_SUCCESS;
  if (src_dtype == QMG_C32) k_to_half<float><<<grid_1d(n), BLOCK, 0, as_stream(stream)>>>((__half2*)dst_c16, src, (long)n);
  else k_to_half<double><<<grid_1d(n), BLOCK, 0, as_stream(stream)>>>((__half2*)dst_c16, src, (long)n);
  QMG_LAUNCH_CHECK();
  return QMG_SUCCESS;
}

// complex<half> -> complex<double> or complex<float> (dst_dtype), exact
int qmg_convert_from_c16(void* dst, int dst_dtype, const void* src_c16, size_t n, void* stream) {
  if (!valid_dtype(dst_dtype) || ((!dst || !src_c16) && n)) return QMG_ERR_INVALID;
  if (n == 0) return QMG_SUCCESS;
  if (dst_dtype == QMG_C32) k_from_half<float><<<grid_1d(n), BLOCK, 0, as_stream(stream)>>>(dst, (const __half2*)src_c16, (long)n);
  else k_from_half<double><<<grid_1d(n), BLOCK, 0, as_stream(stream)>>>(dst, (const __half2*)src_c16, (long)n);
  QMG_LAUNCH_CHECK();
  return QMG_SUCCESS;
}

// lhs (+)= pieces(M) rhs with d->clover / d->hopping stored as complex<half>, vectors complex<float>, fp32 arithmetic.
// nc = 2 only (QMG_ERR_UNSUPPORTED otherwise); nrhs <= 16 with an active mask, as qmg_stencil_apply_t.
int qmg_stencil_apply_h16(const qmg_stencil_desc* d, void* lhs, const void* rhs, unsigned pieces, int nrhs, size_t vec_stride, unsigned mask, void* stream) {
  if (!d || !lhs || !rhs || nrhs < 1 || nrhs > 16) return QMG_ERR_INVALID;
  if (!valid_lattice(d->Lx, d->Ly)) return QMG_ERR_INVALID;
  if (d->nc != 2) return QMG_ERR_UNSUPPORTED;
  if (nrhs > 1 && vec_stride < (size_t)d->Lx * d->Ly * 2) return QMG_ERR_INVALID;
  unsigned char ridx[16];
  int n = 0;
  for (int k = 0; k < nrhs; k++)
    if ((mask >> k) & 1u) ridx[n++] = (unsigned char)k;
  if (n == 0) return QMG_SUCCESS;
  return site_kernel_apply(0, d, lhs, rhs, pieces, n, (long)vec_stride, ridx, as_stream(stream), false, nullptr);
}

// One y-slab of a lattice: lhs (+)= pieces(M) rhs on the slab's rows, the right-hand side's rows -1 / Ly taken from
// halo_lo / halo_hi (filled by qmg_halo_exchange).  storage: QMG_C64, QMG_C32, or QMG_C32 | QMG_SLAB_H16 for 16-bit stored
// matrices.  rows: 0 = all, 1 = interior rows only (need no halo: they can run while the exchange is in flight),
// 2 = the two boundary rows.  nc = 2: kernel S in any storage and any `rows`; other nc (the Galerkin coarse operators): kernel B,
// fp64 or fp32, rows = 0 (QMG_ERR_UNSUPPORTED otherwise).
int qmg_stencil_apply_slab(int storage, const qmg_stencil_desc* d, void* lhs, const void* rhs, const void* halo_lo, const void* halo_hi,
                           unsigned pieces, int nrhs, size_t vec_stride, size_t halo_stride, unsigned mask, int rows, void* stream) {
  if (!d || !lhs || !rhs || !halo_lo || !halo_hi || nrhs < 1 || nrhs > 16 || rows < 0 || rows > 2) return QMG_ERR_INVALID;
  if (lhs == rhs) {   // in place only for the reference's aliased use (stencil_2d.h:1904): ONE parity written, from hops alone
    const unsigned ev = pieces & (QMG_P_CLOVER_E | QMG_P_EO | QMG_P_SHIFT_E | QMG_P_ZERO_E), od = pieces & (QMG_P_CLOVER_O | QMG_P_OE | QMG_P_SHIFT_O | QMG_P_ZERO_O);
    if ((ev && od) || (pieces & (QMG_P_CLOVER | QMG_P_SHIFT))) return QMG_ERR_INVALID;
  }
  if (!valid_lattice(d->Lx, d->Ly)) return QMG_ERR_INVALID;
  const bool h16 = storage & QMG_SLAB_H16, m32 = storage & QMG_SLAB_M32, m16 = storage & QMG_SLAB_M16;
  const int dtype = storage & ~(QMG_SLAB_H16 | QMG_SLAB_M32 | QMG_SLAB_M16);
  if (!valid_dtype(dtype) || (h16 && dtype != QMG_C32) || (m32 && m16) || ((m32 || m16) && d->nc == 2)) return QMG_ERR_INVALID;
  if (nrhs > 1 && (vec_stride < (size_t)d->Lx * d->Ly * d->nc || halo_stride < (size_t)d->Lx * d->nc)) return QMG_ERR_INVALID;
  unsigned char ridx[16];
  int n = 0;
  for (int k = 0; k < nrhs; k++)
    if ((mask >> k) & 1u) ridx[n++] = (unsigned char)k;
  if (n == 0) return QMG_SUCCESS;
  SlabHalo slab;
  slab.lo = halo_lo; slab.hi = halo_hi; slab.stride = (long)halo_stride; slab.rows = rows;
  if (d->nc != 2) {   // any other nc: kernels B / B32 / C (all rows in one launch); matrices in the vectors' precision or narrower
    if (rows != 0) return QMG_ERR_UNSUPPORTED;
    const int vec32 = dtype == QMG_C32 ? 1 : 0;
    const int mat = (h16 || m16) ? 2 : (m32 || vec32) ? 1 : 0;
    if ((mat == 2 && (d->nc & 3)) || ((mat == 2 || m32) && d->nc <= 4)) return QMG_ERR_UNSUPPORTED;   // (narrow storage: the Galerkin levels, nc > 4)
    return generic_slab_apply(d, lhs, rhs, pieces, n, (long)vec_stride, ridx, as_stream(stream), &slab, mat, vec32);
  }
  return site_kernel_apply(h16 ? 0 : (dtype == QMG_C32 ? 1 : 2), d, lhs, rhs, pieces, n, (long)vec_stride, ridx, as_stream(stream), false, &slab);
}

}  // extern "C"

// qmg_wilson.hip -- the Wilson operator applied STRAIGHT FROM THE GAUGE LINKS (kernel W): no stored stencil matrices.
//
// The reference's Wilson2D fills a general 2x2-matrix stencil from the U(1) links (operators/wilson.h:153-209) and applies it
// with the general stencil code: 5 matrices of 64 B + 2 vectors of 32 B = 384 B/site in fp64.  Every one of those matrices
// is a fixed 2x2 spin pattern times ONE link, so they can be rebuilt in registers from the four links a site touches:
//   +x: 1/2 [[-w, 1],[ 1,-w]] Ux(x)        -x: 1/2 [[-w,-1],[-1,-w]] conj Ux(x - x^)
//   +y: 1/2 [[-w,-i],[ i,-w]] Uy(x)        -y: 1/2 [[-w, i],[-i,-w]] conj Uy(x - y^)        clover: 2w on the diagonal
// Bytes: every link is used by the two sites it joins, so the links cost 2 x 16 B/site (the second use is an L2 hit, like
// the neighbours of the right-hand side) -- 32 + 32 + 32 = 96 B/site in fp64 (1/4 of the stored stencil), 48 B/site in fp32.
// The arithmetic is the stored path's: the matrix entries are formed by the same single multiplications k_wilson_fill does
// and enter the same FMA sequence as the site kernel (qmg_site.hip: per-column partial sums, added at the end), so in fp64
// the result is BIT-IDENTICAL to qmg_stencil_apply on the filled stencil through that kernel (tests/test_gpu_wilson_direct.py).
//
// Layout: fp64 -- two lanes per site, lane c holds component c of every site vector it needs (one 16-byte chunk each,
// perfectly coalesced), accumulates column c's contribution to both output rows and swaps one complex number with its
// partner (DPP); fp32 -- one lane per site (the site vector is one 16-byte chunk).  Links: own Ux, Uy and the -x / -y
// neighbours' (opposite parity).  Load phase as raw registers behind a scheduling barrier, as in qmg_site.hip.
// Piece sets served: clover + all hops (+ shifts) of the processed parities (apply_M and its one-parity forms) and hops
// only (D_eo / D_oe); anything else returns QMG_ERR_UNSUPPORTED and the caller uses the stored stencil.
// The hops of the RIGHT-BLOCK-JACOBI stencil (stencil_2d.h:1556-1581: H'_mu(x) = H_mu(x) . cinv(x + mu)) come from the links
// too when cinv is one real number times the identity -- Wilson with a real mass and no eo / dof shift: cinv = 1 / (2w + m) --:
// every entry is then the stored entry times that number (SHAPE 3, qmg_wilson_hops_direct), which is exactly what the build's
// 2x2 product leaves in memory, so the Schur-complement applies of the K-cycle stream 128 instead of 320 B per written site.
// y-slabs: the links are indexed on the GLOBAL lattice (replicated, 32 B/site), only the right-hand side's rows
// -1 / Ly come from halo buffers.
#include <type_traits>

#include "qmg_common.h"

namespace qmg {

int g_wilson_pair = 2;   // tuning knob "wilson_pair": the full operator through the paired-parity kernel W2 (1), on two rows per lane group where that applies (2)

struct WilsonArgs {
  const void* gauge;       // [mu][global site] complex<T>
  void* lhs;
  const void* rhs;
  int hr, Ly;              // the vectors' lattice (a slab: its local rows)
  int gLy, gy0;            // the gauge field's lattice and the slab's first row on it
  long half_vol, ghalf_vol;
  unsigned pieces;
  int nrhs;
  long vec_stride;
  int par_first, par_count, nrows;
  double w;
  double hop_scale;        // SHAPE 3: every hop entry times this (the right-block-Jacobi hops of a uniform cinv)
  double shift[2], eo_shift[2], dof_shift[2];
  int ridx[16];
  const void* halo_lo;
  const void* halo_hi;
  long halo_stride;
  int y_first, y_count, boundary_only;
  Epilogue epi;            // out = other_scale other + acc_scale acc, MR dots of out (qmg_common.h); one system per launch
};

typedef float w4f __attribute__((ext_vector_type(4)));
typedef float w2f __attribute__((ext_vector_type(2)));
typedef double w2d __attribute__((ext_vector_type(2)));

template <typename R>
__device__ __forceinline__ void fmac2w(R& ax, R& ay, R mx, R my, R bx, R by) {
  ax = fma(mx, bx, ax); ax = fma(-my, by, ax);
  ay = fma(mx, by, ay); ay = fma(my, bx, ay);
}

// a row-uniform pointer, told to the compiler: the loads then take the scalar-base + 32-bit-offset form
// (the result is a GLOBAL-address-space pointer: an integer cast to a plain pointer would make the accesses flat_load / flat_store)
typedef __attribute__((address_space(1))) char gchar;
__device__ __forceinline__ gchar* uni(const char* p) {
  const unsigned long long v = reinterpret_cast<unsigned long long>(p);
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
  return (gchar*)(((unsigned long long)hi << 32) | lo);
}
template <typename V> __device__ __forceinline__ V gld(const gchar* base, unsigned off) { return *(const __attribute__((address_space(1))) V*)(base + off); }
template <typename V> __device__ __forceinline__ V gld_nt(const gchar* base, unsigned off) {
  return __builtin_nontemporal_load((const __attribute__((address_space(1))) V*)(base + off));
}

// the four hop matrices' entries [row][col] for column c from the links (k_wilson_fill's multiplications, qmg_fill.hip)
template <typename R>
struct HopCol { R m0x, m0y, m1x, m1y; };   // entry (row 0, col c) and (row 1, col c)
template <typename R, bool SCALED>
__device__ __forceinline__ HopCol<R> hop_col(int d, int c, R ux, R uy, R hw, R sc) {
  // (ux, uy) = the link (already conjugated for the backward directions)
  const R dx = hw * ux, dy = hw * uy;                 // diagonal entry: -w/2 U
  const R h = (R)0.5;
  R ox, oy;                                           // off-diagonal entry (row != col)
  HopCol<R> o;
  if (d == 0) { ox = h * ux; oy = h * uy; }                                   // +x: U/2 both
  else if (d == 2) { ox = -h * ux; oy = -h * uy; }                            // -x: -U/2 both
  else if (d == 1) {                                                          // +y: [0][1] = -iU/2, [1][0] = iU/2
    if (c == 1) { ox = h * uy; oy = -h * ux; } else { ox = -h * uy; oy = h * ux; }
  } else {                                                                    // -y: [0][1] = iU/2, [1][0] = -iU/2
    if (c == 1) { ox = -h * uy; oy = h * ux; } else { ox = h * uy; oy = -h * ux; }
  }
  if (c == 0) { o.m0x = dx; o.m0y = dy; o.m1x = ox; o.m1y = oy; }
  else { o.m0x = ox; o.m0y = oy; o.m1x = dx; o.m1y = dy; }
  if (SCALED) { o.m0x *= sc; o.m0y *= sc; o.m1x *= sc; o.m1y *= sc; }   // the stored entry times cinv (k_rb_hopping's product)
  return o;
}

// this lane's 16-byte chunk of an epilogue vector at byte offset boff from its base (or zeros when the vector is absent)
__device__ __forceinline__ w4f epi_chunk(const void* vec, long boff) {
  w4f z; z.x = 0.f; z.y = 0.f; z.z = 0.f; z.w = 0.f;
  return vec ? *reinterpret_cast<const w4f*>(reinterpret_cast<const char*>(vec) + boff) : z;
}

// One site's (fp64: one site's column c0) result from its five right-hand-side chunks xr = {+x, +y, -x, -y, own} and its four
// links (backward ones already conjugated), then the store.  The order of operations is kernel S's (qmg_site.hip).
// SHAPE 1: clover + hops (+ shift); 2: hops; 3: hops scaled by a.hop_scale.
// live = false: a padding lane past the end of the half row (it computed a copy of the last site so that every lane of the wavefront
// stays active for the epilogue's wavefront sums): nothing is stored, nothing is added to the dots.  ec: this lane's chunk of the
// epilogue's vector, loaded by the caller IN ITS LOAD PHASE (a load issued here, after the arithmetic, would add one full memory
// latency to every wavefront: measured +4 % on the whole K-cycle).
// EPI is a compile-time mode: the launches without an epilogue keep their register count (a run-time branch cost pair2<float> 34 VGPRs and a
// wavefront of occupancy), and each mode pays only for what it loads:
//   0 none | 1 dots against the right-hand side's own-site chunk (MR on the full operator: p = A r, <p,r>: nothing extra is loaded) |
//   2 out = os other + as acc | 3 the same, and dots against `other` (the Schur complement's r_e - D_eo t with MR dots against r_e) |
//   4 dots against a separate vector.  Modes 2-4 load ONE extra chunk per site (`ec`).
template <typename T, int SHAPE, int EPI>
__device__ __forceinline__ void wilson_site(const w4f (&xr)[5], const T (&lx)[4], const T (&ly)[4], T hw, T cw, bool do_shift, bool do_zero, int p, int c0,
                                            const WilsonArgs& a, gchar* dst_chunk, const w4f& ec, bool live, double (&ed)[3]) {
  constexpr bool F64 = sizeof(T) == 8;
  constexpr int NCOL = F64 ? 1 : 2;
  typedef T R;
  // acc[col][row]: column `col`'s contribution to output row `row` (the site kernel's per-lane partial sums)
  R ax[NCOL][2], ay[NCOL][2];
#pragma unroll
  for (int cc = 0; cc < NCOL; cc++) { ax[cc][0] = ax[cc][1] = ay[cc][0] = ay[cc][1] = (R)0; }
#pragma unroll
  for (int cc = 0; cc < NCOL; cc++) {
    const int c = F64 ? c0 : cc;
    // component c of a chunk: fp64 chunk = that component; fp32 chunk = (x0.re, x0.im, x1.re, x1.im)
    auto comp = [&](const w4f& v, R& vx, R& vy) {
      if (F64) { const w2d q = __builtin_bit_cast(w2d, v); vx = (R)q.x; vy = (R)q.y; }
      else { vx = (R)(c ? v.z : v.x); vy = (R)(c ? v.w : v.y); }
    };
    R vx, vy;
    if (SHAPE == 1) {   // clover first: 2w on the diagonal (row == col)
      comp(xr[4], vx, vy);
      fmac2w<R>(ax[cc][c], ay[cc][c], cw, (R)0, vx, vy);
    }
#pragma unroll
    for (int d = 0; d < 4; d++) {
      comp(xr[d], vx, vy);
      const HopCol<R> m = hop_col<R, SHAPE == 3>(d, c, lx[d], ly[d], hw, (R)a.hop_scale);
      fmac2w<R>(ax[cc][0], ay[cc][0], m.m0x, m.m0y, vx, vy);
      fmac2w<R>(ax[cc][1], ay[cc][1], m.m1x, m.m1y, vx, vy);
    }
    if (do_shift) {   // shift +- eo_shift +- dof_shift on the diagonal (stencil_2d.h:890-908)
      const double sg = p ? -1.0 : 1.0, dg = c ? -1.0 : 1.0;
      const R sx = (R)(a.shift[0] + sg * a.eo_shift[0] + dg * a.dof_shift[0]), sy = (R)(a.shift[1] + sg * a.eo_shift[1] + dg * a.dof_shift[1]);
      comp(xr[4], vx, vy);
      fmac2w<R>(ax[cc][c], ay[cc][c], sx, sy, vx, vy);
    }
  }
  if (F64) {
    // lane c keeps row c: own column's part + the partner's part of that row
    const int c = c0;
    const double sendx = c ? (double)ax[0][0] : (double)ax[0][1], sendy = c ? (double)ay[0][0] : (double)ay[0][1];
    const double recvx = lane_xor1(sendx), recvy = lane_xor1(sendy);
    const double ownx = c ? (double)ax[0][1] : (double)ax[0][0], owny = c ? (double)ay[0][1] : (double)ay[0][0];
    w2d o;
    o.x = c ? (recvx + ownx) : (ownx + recvx);
    o.y = c ? (recvy + owny) : (owny + recvy);
    __attribute__((address_space(1))) w2d* dst = (__attribute__((address_space(1))) w2d*)dst_chunk;
    if (!do_zero) { const w2d pv = *dst; o.x += pv.x; o.y += pv.y; }
    if (EPI) {
      if (EPI == 2 || EPI == 3) {
        const w2d ov = __builtin_bit_cast(w2d, ec);
        o.x = fma(a.epi.other_scale, ov.x, a.epi.acc_scale * o.x); o.y = fma(a.epi.other_scale, ov.y, a.epi.acc_scale * o.y);
      } else if (a.epi.acc_scale != 1.0) { o.x *= a.epi.acc_scale; o.y *= a.epi.acc_scale; }
      if (EPI != 2 && live) {
        const w2d r = __builtin_bit_cast(w2d, EPI == 1 ? xr[4] : ec);
        ed[0] = fma(r.x, o.x, ed[0]); ed[0] = fma(r.y, o.y, ed[0]);
        ed[1] = fma(r.x, o.y, ed[1]); ed[1] = fma(-r.y, o.x, ed[1]);
        ed[2] = fma(o.x, o.x, ed[2]); ed[2] = fma(o.y, o.y, ed[2]);
      }
    }
    if (live) __builtin_nontemporal_store(o, dst);
  } else {
    w4f o;
    o.x = (float)(ax[0][0] + ax[NCOL - 1][0]); o.y = (float)(ay[0][0] + ay[NCOL - 1][0]);
    o.z = (float)(ax[0][1] + ax[NCOL - 1][1]); o.w = (float)(ay[0][1] + ay[NCOL - 1][1]);
    __attribute__((address_space(1))) w4f* dst = (__attribute__((address_space(1))) w4f*)dst_chunk;
    if (!do_zero) { const w4f pv = *dst; o.x += pv.x; o.y += pv.y; o.z += pv.z; o.w += pv.w; }
    if (EPI) {
      if (EPI == 2 || EPI == 3) {
        const w4f ov = ec;
        const float os = (float)a.epi.other_scale, as = (float)a.epi.acc_scale;
        o.x = fmaf(os, ov.x, as * o.x); o.y = fmaf(os, ov.y, as * o.y); o.z = fmaf(os, ov.z, as * o.z); o.w = fmaf(os, ov.w, as * o.w);
      } else if (a.epi.acc_scale != 1.0) { const float as = (float)a.epi.acc_scale; o.x *= as; o.y *= as; o.z *= as; o.w *= as; }
      if (EPI != 2 && live) {   // both components of the site, accumulated in fp64 from the values as stored
        const w4f r = (EPI == 1) ? xr[4] : ec;
        const double r0x = r.x, r0y = r.y, r1x = r.z, r1y = r.w, o0x = o.x, o0y = o.y, o1x = o.z, o1y = o.w;
        ed[0] = fma(r0x, o0x, ed[0]); ed[0] = fma(r0y, o0y, ed[0]); ed[0] = fma(r1x, o1x, ed[0]); ed[0] = fma(r1y, o1y, ed[0]);
        ed[1] = fma(r0x, o0y, ed[1]); ed[1] = fma(-r0y, o0x, ed[1]); ed[1] = fma(r1x, o1y, ed[1]); ed[1] = fma(-r1y, o1x, ed[1]);
        ed[2] = fma(o0x, o0x, ed[2]); ed[2] = fma(o0y, o0y, ed[2]); ed[2] = fma(o1x, o1x, ed[2]); ed[2] = fma(o1y, o1y, ed[2]);
      }
    }
    if (live) __builtin_nontemporal_store(o, dst);
  }
}

// end of a kernel W launch with the MR epilogue: one partial per wavefront of the launch (system slot 0); EVERY lane calls it
__device__ __forceinline__ void wilson_store_partials(const WilsonArgs& a, double (&ed)[3]) {
  const double s0 = wave_sum(ed[0]), s1 = wave_sum(ed[1]), s2 = wave_sum(ed[2]);
  if ((threadIdx.x & (WAVE - 1)) == 0) {
    const long w = ((long)blockIdx.y * gridDim.x + blockIdx.x) * (BLOCK / WAVE) + threadIdx.x / WAVE;
    double* pp = a.epi.part + w * 4;
    pp[0] = s0; pp[1] = s1; pp[2] = s2; pp[3] = 0.0;
  }
}

// T = storage scalar (double: 2 lanes per site; float: 1 lane per site).  SHAPE 1: clover + hops (+ shift); 2: hops only;
// 3: hops only, entries scaled (right-block-Jacobi).
template <typename T, int SHAPE, bool ZERO, bool BATCH, int EPI = 0>
__global__ __launch_bounds__(BLOCK) void k_wilson_direct(const WilsonArgs a) {
  constexpr bool F64 = sizeof(T) == 8;
  constexpr int LPS = F64 ? 2 : 1;
  typedef T R;
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  const int jt = t / LPS, c0 = F64 ? (t % LPS) : 0;
  if (!EPI && jt >= a.hr) return;
  const bool live = !EPI || jt < a.hr;
  const int j = live ? jt : a.hr - 1;   // with an epilogue padding lanes shadow the last site (see wilson_site); without one they have left
  double ed[3] = {0.0, 0.0, 0.0};
  const long sys_bytes = a.vec_stride * (long)(2 * sizeof(T));
  const R hw = (R)(-0.5 * a.w), cw = (R)(2.0 * a.w);
  for (int row = blockIdx.y; row < a.nrows; row += gridDim.y) {
    const int p = (a.par_count == 2) ? (row & 1) : a.par_first;
    const int yi = (a.par_count == 2) ? (row >> 1) : row;
    const int y = a.boundary_only ? (yi ? a.Ly - 1 : 0) : a.y_first + yi;
    const bool do_shift = (a.pieces >> (10 + p)) & 1u;
    const bool do_zero = ZERO || ((a.pieces >> (12 + p)) & 1u);
    const int s = (y + p) & 1;
    int jp = j + s;     if (jp == a.hr) jp = 0;
    int jm = j + s - 1; if (jm < 0) jm = a.hr - 1;
    const int yp = (y + 1 == a.Ly) ? 0 : y + 1;
    const int ym = (y == 0) ? a.Ly - 1 : y - 1;
    // Addresses = a row-uniform 64-bit base (SGPRs) + a 32-bit lane offset: the loads take the scalar-base form and an
    // address costs one VGPR instead of two (a half-row is far below 4 GiB).
    constexpr unsigned CH = 16u * LPS;                             // bytes of a site vector
    const unsigned off_j = (unsigned)j * CH + (unsigned)c0 * 16u, off_jp = (unsigned)jp * CH + (unsigned)c0 * 16u, off_jm = (unsigned)jm * CH + (unsigned)c0 * 16u;
    const long row_own = ((long)p * a.half_vol + (long)y * a.hr) * CH;
    const long opp = (long)(1 - p) * a.half_vol;
    const long row_x = (opp + (long)y * a.hr) * CH, row_yp = (opp + (long)yp * a.hr) * CH, row_ym = (opp + (long)ym * a.hr) * CH;
    const long hrow = (long)(1 - p) * a.hr * CH;                   // the halo buffers' row of the opposite parity
    // links on the global lattice
    const int gy = a.gy0 + y;
    const int gym = (gy == 0) ? a.gLy - 1 : gy - 1;
    const long gvol = 2 * a.ghalf_vol;
    constexpr unsigned GB = 2u * sizeof(T);                        // bytes of a link
    const char* gc = reinterpret_cast<const char*>(a.gauge);
    const gchar* g_own_x = uni(gc + ((long)p * a.ghalf_vol + (long)gy * a.hr) * GB);
    const gchar* g_own_y = uni(gc + (gvol + (long)p * a.ghalf_vol + (long)gy * a.hr) * GB);
    const gchar* g_xm = uni(gc + ((long)(1 - p) * a.ghalf_vol + (long)gy * a.hr) * GB);
    const gchar* g_ym = uni(gc + (gvol + (long)(1 - p) * a.ghalf_vol + (long)gym * a.hr) * GB);
    const unsigned goff_j = (unsigned)j * GB, goff_jm = (unsigned)jm * GB;
    // ---- load phase: first system's chunks, then the four links (raw)
    const bool from_hi = a.halo_hi && y + 1 == a.Ly, from_lo = a.halo_lo && y == 0;
    w4f xr[5];                                     // neighbours +x +y -x -y, own
    auto load_x = [&](int k) {
      const long off = (long)a.ridx[k] * sys_bytes;
      const char* x = reinterpret_cast<const char*>(a.rhs) + off;
      const long hoff = (long)a.ridx[k] * a.halo_stride * (long)(2 * sizeof(T));
      const gchar* b1 = uni(from_hi ? reinterpret_cast<const char*>(a.halo_hi) + hoff + hrow : x + row_yp);
      const gchar* b3 = uni(from_lo ? reinterpret_cast<const char*>(a.halo_lo) + hoff + hrow : x + row_ym);
      const gchar* bx = uni(x + row_x);
      xr[0] = gld<w4f>(bx, off_jp);
      xr[1] = gld<w4f>(b1, off_j);
      xr[2] = gld<w4f>(bx, off_jm);
      xr[3] = gld<w4f>(b3, off_j);
      if (SHAPE == 1 || do_shift) xr[4] = gld<w4f>(uni(x + row_own), off_j);
    };
    load_x(0);
    w4f e_c = xr[0];                               // the epilogue's chunk of this site (modes 2-4; a placeholder otherwise)
    if (EPI >= 2) e_c = epi_chunk(EPI == 4 ? a.epi.dotv : a.epi.other, (long)a.ridx[0] * sys_bytes + row_own + off_j);
    R lx[4], ly[4];                                // links per direction, conjugated for the backward ones
    if (F64) {
      const w2d u0 = gld_nt<w2d>(g_own_x, goff_j);
      const w2d u1 = gld_nt<w2d>(g_own_y, goff_j);
      const w2d u2 = gld<w2d>(g_xm, goff_jm);
      const w2d u3 = gld<w2d>(g_ym, goff_j);
      __builtin_amdgcn_sched_barrier(0);
      lx[0] = (R)u0.x; ly[0] = (R)u0.y; lx[1] = (R)u1.x; ly[1] = (R)u1.y;
      lx[2] = (R)u2.x; ly[2] = -(R)u2.y; lx[3] = (R)u3.x; ly[3] = -(R)u3.y;
    } else {
      const w2f u0 = gld_nt<w2f>(g_own_x, goff_j);
      const w2f u1 = gld_nt<w2f>(g_own_y, goff_j);
      const w2f u2 = gld<w2f>(g_xm, goff_jm);
      const w2f u3 = gld<w2f>(g_ym, goff_j);
      __builtin_amdgcn_sched_barrier(0);
      lx[0] = (R)u0.x; ly[0] = (R)u0.y; lx[1] = (R)u1.x; ly[1] = (R)u1.y;
      lx[2] = (R)u2.x; ly[2] = -(R)u2.y; lx[3] = (R)u3.x; ly[3] = -(R)u3.y;
    }
    const int nsys = BATCH ? a.nrhs : 1;
    for (int k = 0; k < nsys; k++) {
      if (BATCH && k > 0) { load_x(k); __builtin_amdgcn_sched_barrier(0); }
      char* out = reinterpret_cast<char*>(a.lhs) + (long)a.ridx[k] * sys_bytes;
      wilson_site<T, SHAPE, EPI>(xr, lx, ly, hw, cw, do_shift, do_zero, p, c0, a, uni(out + row_own) + off_j, e_c, live, ed);
    }
  }
  if (EPI && EPI != 2) wilson_store_partials(a, ed);
}

// Kernel W2: BOTH parities of column j on row y per lane group -- the full operator (clover + all hops on both parities).
// Kernel W is not HBM-bound but in-flight-bound (3 KB of HBM requests per wavefront); a pair shares what the two sites
// have in common -- each site's own chunk is an x-neighbour of the other, one back link is the other's own link: 15 loads
// per pair instead of 18 -- and puts twice the HBM bytes of a wavefront in flight.  Per-site arithmetic = wilson_site.
template <typename T, bool ZERO, bool BATCH, int EPI = 0>
__global__ __launch_bounds__(BLOCK) void k_wilson_pair(const WilsonArgs a) {
  constexpr bool F64 = sizeof(T) == 8;
  constexpr int LPS = F64 ? 2 : 1;
  typedef T R;
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  const int jt = t / LPS, c0 = F64 ? (t % LPS) : 0;
  if (!EPI && jt >= a.hr) return;
  const bool live = !EPI || jt < a.hr;
  const int j = live ? jt : a.hr - 1;   // with an epilogue padding lanes shadow the last site (see wilson_site); without one they have left
  double ed[3] = {0.0, 0.0, 0.0};
  const long sys_bytes = a.vec_stride * (long)(2 * sizeof(T));
  const R hw = (R)(-0.5 * a.w), cw = (R)(2.0 * a.w);
  constexpr unsigned CH = 16u * LPS, GB = 2u * sizeof(T);
  const int jl = (j == 0) ? a.hr - 1 : j - 1, jr = (j + 1 == a.hr) ? 0 : j + 1;
  const unsigned off_j = (unsigned)j * CH + (unsigned)c0 * 16u, off_l = (unsigned)jl * CH + (unsigned)c0 * 16u, off_r = (unsigned)jr * CH + (unsigned)c0 * 16u;
  const unsigned goff_j = (unsigned)j * GB, goff_l = (unsigned)jl * GB;
  for (int yi = blockIdx.y; yi < a.y_count; yi += gridDim.y) {
    const int y = a.boundary_only ? (yi ? a.Ly - 1 : 0) : a.y_first + yi;
    const int sE = y & 1;                          // the even site of column j sits at x = 2j + sE, the odd one at 2j + 1 - sE
    const bool shE = (a.pieces >> 10) & 1u, shO = (a.pieces >> 11) & 1u;
    const bool zE = ZERO || ((a.pieces >> 12) & 1u), zO = ZERO || ((a.pieces >> 13) & 1u);
    const int yp = (y + 1 == a.Ly) ? 0 : y + 1, ym = (y == 0) ? a.Ly - 1 : y - 1;
    const long rowE = (long)y * a.hr * CH, rowO = (a.half_vol + (long)y * a.hr) * CH;
    const long rowE_up = (long)yp * a.hr * CH, rowE_dn = (long)ym * a.hr * CH;
    const long rowO_up = (a.half_vol + (long)yp * a.hr) * CH, rowO_dn = (a.half_vol + (long)ym * a.hr) * CH;
    const bool from_hi = a.halo_hi && y + 1 == a.Ly, from_lo = a.halo_lo && y == 0;
    // the x-neighbour each site does not get from its partner: sE = 0: E's -x (odd row, j-1) and O's +x (even row, j+1); sE = 1: mirrored
    const unsigned off_oth_O = sE ? off_r : off_l;      // in the ODD row, for the even site
    const unsigned off_oth_E = sE ? off_l : off_r;      // in the EVEN row, for the odd site
    w4f ownE, ownO, othO, othE, upE, dnE, upO, dnO;     // upE / dnE: the even site's +y / -y neighbours (odd rows), etc.
    auto load_x = [&](int k) {
      const long off = (long)a.ridx[k] * sys_bytes;
      const char* x = reinterpret_cast<const char*>(a.rhs) + off;
      const long hoff = (long)a.ridx[k] * a.halo_stride * (long)(2 * sizeof(T));
      const gchar* bE = uni(x + rowE);
      const gchar* bO = uni(x + rowO);
      const long hE = 0, hO = (long)a.hr * CH;          // halo buffers: [parity][hr] site vectors
      const gchar* bO_up = uni(from_hi ? reinterpret_cast<const char*>(a.halo_hi) + hoff + hO : x + rowO_up);
      const gchar* bE_up = uni(from_hi ? reinterpret_cast<const char*>(a.halo_hi) + hoff + hE : x + rowE_up);
      const gchar* bO_dn = uni(from_lo ? reinterpret_cast<const char*>(a.halo_lo) + hoff + hO : x + rowO_dn);
      const gchar* bE_dn = uni(from_lo ? reinterpret_cast<const char*>(a.halo_lo) + hoff + hE : x + rowE_dn);
      ownE = gld<w4f>(bE, off_j); ownO = gld<w4f>(bO, off_j);
      othO = gld<w4f>(bO, off_oth_O); othE = gld<w4f>(bE, off_oth_E);
      upE = gld<w4f>(bO_up, off_j); dnE = gld<w4f>(bO_dn, off_j);
      upO = gld<w4f>(bE_up, off_j); dnO = gld<w4f>(bE_dn, off_j);
    };
    load_x(0);
    w4f ecE = ownE, ecO = ownO;   // the epilogue's chunks of the two sites (modes 2-4)
    if (EPI >= 2) {
      const void* ev = (EPI == 4) ? a.epi.dotv : a.epi.other;
      ecE = epi_chunk(ev, (long)a.ridx[0] * sys_bytes + rowE + off_j); ecO = epi_chunk(ev, (long)a.ridx[0] * sys_bytes + rowO + off_j);
    }
    // links (global lattice): own of both sites, the one back-x link that is not the partner's own, two back-y links
    const int gy = a.gy0 + y;
    const int gym = (gy == 0) ? a.gLy - 1 : gy - 1;
    const long gvol = 2 * a.ghalf_vol;
    const char* gc = reinterpret_cast<const char*>(a.gauge);
    const gchar* gxE = uni(gc + ((long)gy * a.hr) * GB);
    const gchar* gxO = uni(gc + (a.ghalf_vol + (long)gy * a.hr) * GB);
    const gchar* gyE = uni(gc + (gvol + (long)gy * a.hr) * GB);
    const gchar* gyO = uni(gc + (gvol + a.ghalf_vol + (long)gy * a.hr) * GB);
    const gchar* gyE_dn = uni(gc + (gvol + (long)gym * a.hr) * GB);                 // Uy of the EVEN sites of row y-1: the odd site's back-y link
    const gchar* gyO_dn = uni(gc + (gvol + a.ghalf_vol + (long)gym * a.hr) * GB);   // ... of the ODD sites: the even site's
    typedef typename std::conditional<F64, w2d, w2f>::type LK;
    const LK uxE = gld<LK>(gxE, goff_j), uxO = gld<LK>(gxO, goff_j), uyE = gld<LK>(gyE, goff_j), uyO = gld<LK>(gyO, goff_j);
    const LK ubx = gld<LK>(sE ? gxE : gxO, goff_l);     // sE = 0: Ux of the odd site at j-1 (the even site's back-x); sE = 1: Ux of the even site at j-1
    const LK ubyE = gld<LK>(gyO_dn, goff_j), ubyO = gld<LK>(gyE_dn, goff_j);
    __builtin_amdgcn_sched_barrier(0);
    const int nsys = BATCH ? a.nrhs : 1;
    for (int k = 0; k < nsys; k++) {
      if (BATCH && k > 0) { load_x(k); __builtin_amdgcn_sched_barrier(0); }
      char* out = reinterpret_cast<char*>(a.lhs) + (long)a.ridx[k] * sys_bytes;
      // even site: +x = odd row at j + sE, -x = odd row at j + sE - 1; odd site: +x = even row at j + 1 - sE, -x = even row at j - sE
      {   // (the links are unpacked site by site, right before use: fewer live registers than both sets up front)
        const LK bxE = sE ? uxO : ubx;                   // Ux at the even site's -x neighbour (an odd site at jmE = j + sE - 1)
        const R lxE[4] = {(R)uxE.x, (R)uyE.x, (R)bxE.x, (R)ubyE.x}, lyE[4] = {(R)uxE.y, (R)uyE.y, -(R)bxE.y, -(R)ubyE.y};
        const w4f xrE[5] = {sE ? othO : ownO, upE, sE ? ownO : othO, dnE, ownE};
        wilson_site<T, 1, EPI>(xrE, lxE, lyE, hw, cw, shE, zE, 0, c0, a, uni(out + rowE) + off_j, ecE, live, ed);
      }
      {
        const LK bxO = sE ? ubx : uxE;                   // Ux at the odd site's -x neighbour (an even site at jmO = j - sE)
        const R lxO[4] = {(R)uxO.x, (R)uyO.x, (R)bxO.x, (R)ubyO.x}, lyO[4] = {(R)uxO.y, (R)uyO.y, -(R)bxO.y, -(R)ubyO.y};
        const w4f xrO[5] = {sE ? ownE : othE, upO, sE ? othE : ownE, dnO, ownO};
        wilson_site<T, 1, EPI>(xrO, lxO, lyO, hw, cw, shO, zO, 1, c0, a, uni(out + rowO) + off_j, ecO, live, ed);
      }
    }
  }
  if (EPI && EPI != 2) wilson_store_partials(a, ed);
}

// Kernel W2 on TWO consecutive rows per lane group (one system): rows y and y + 1 are each other's +-y neighbours and back-y
// links, so the four sites take 12 + 12 loads instead of 2 x (8 + 7), and -- what matters for a kernel that is bound by the
// bytes a wavefront has in flight -- every wavefront requests 12 KB of HBM data instead of 6 (at 3 resident wavefronts per SIMD
// instead of 4).  Per-site arithmetic = wilson_site, so the results are the one-row kernel's bit for bit.
template <typename T, bool ZERO, int EPI = 0>
__global__ __launch_bounds__(BLOCK) void k_wilson_pair2(const WilsonArgs a) {
  constexpr bool F64 = sizeof(T) == 8;
  constexpr int LPS = F64 ? 2 : 1;
  typedef T R;
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  const int jt = t / LPS, c0 = F64 ? (t % LPS) : 0;
  if (!EPI && jt >= a.hr) return;
  const bool live = !EPI || jt < a.hr;
  const int j = live ? jt : a.hr - 1;   // with an epilogue padding lanes shadow the last site (see wilson_site); without one they have left
  double ed[3] = {0.0, 0.0, 0.0};
  const long sys_bytes = a.vec_stride * (long)(2 * sizeof(T));
  const R hw = (R)(-0.5 * a.w), cw = (R)(2.0 * a.w);
  constexpr unsigned CH = 16u * LPS, GB = 2u * sizeof(T);
  const int jl = (j == 0) ? a.hr - 1 : j - 1, jr = (j + 1 == a.hr) ? 0 : j + 1;
  const unsigned off_j = (unsigned)j * CH + (unsigned)c0 * 16u, off_l = (unsigned)jl * CH + (unsigned)c0 * 16u, off_r = (unsigned)jr * CH + (unsigned)c0 * 16u;
  const unsigned goff_j = (unsigned)j * GB, goff_l = (unsigned)jl * GB;
  const bool shE = (a.pieces >> 10) & 1u, shO = (a.pieces >> 11) & 1u;
  const bool zE = ZERO || ((a.pieces >> 12) & 1u), zO = ZERO || ((a.pieces >> 13) & 1u);
  typedef typename std::conditional<F64, w2d, w2f>::type LK;
  for (int yi = 2 * blockIdx.y; yi < a.y_count; yi += 2 * gridDim.y) {
    const int y = a.y_first + yi;                  // rows y (A) and y + 1 (B); y + 1 <= Ly - 1 (y_count is even)
    const int sA = y & 1, sB = 1 - sA;             // row A: the even site of column j sits at x = 2j + sA
    const int yp2 = (y + 2 == a.Ly) ? 0 : y + 2, ym = (y == 0) ? a.Ly - 1 : y - 1;
    const long rowEA = (long)y * a.hr * CH, rowOA = (a.half_vol + (long)y * a.hr) * CH;
    const long rowEB = rowEA + (long)a.hr * CH, rowOB = rowOA + (long)a.hr * CH;
    const bool from_hi = a.halo_hi && y + 2 == a.Ly, from_lo = a.halo_lo && y == 0;
    const long off = (long)a.ridx[0] * sys_bytes;
    const char* x = reinterpret_cast<const char*>(a.rhs) + off;
    const long hoff = (long)a.ridx[0] * a.halo_stride * (long)(2 * sizeof(T));
    const long hE = 0, hO = (long)a.hr * CH;       // halo buffers: [parity][hr] site vectors
    const gchar* bEA = uni(x + rowEA);
    const gchar* bOA = uni(x + rowOA);
    const gchar* bEB = uni(x + rowEB);
    const gchar* bOB = uni(x + rowOB);
    const gchar* bO_dn = uni(from_lo ? reinterpret_cast<const char*>(a.halo_lo) + hoff + hO : x + (a.half_vol + (long)ym * a.hr) * CH);
    const gchar* bE_dn = uni(from_lo ? reinterpret_cast<const char*>(a.halo_lo) + hoff + hE : x + (long)ym * a.hr * CH);
    const gchar* bO_up = uni(from_hi ? reinterpret_cast<const char*>(a.halo_hi) + hoff + hO : x + (a.half_vol + (long)yp2 * a.hr) * CH);
    const gchar* bE_up = uni(from_hi ? reinterpret_cast<const char*>(a.halo_hi) + hoff + hE : x + (long)yp2 * a.hr * CH);
    // ---- load phase: 12 right-hand-side chunks, 12 links
    const w4f ownEA = gld<w4f>(bEA, off_j), ownOA = gld<w4f>(bOA, off_j), ownEB = gld<w4f>(bEB, off_j), ownOB = gld<w4f>(bOB, off_j);
    const w4f othOA = gld<w4f>(bOA, sA ? off_r : off_l), othEA = gld<w4f>(bEA, sA ? off_l : off_r);
    const w4f othOB = gld<w4f>(bOB, sB ? off_r : off_l), othEB = gld<w4f>(bEB, sB ? off_l : off_r);
    const w4f dnEA = gld<w4f>(bO_dn, off_j), dnOA = gld<w4f>(bE_dn, off_j);      // row y - 1: odd site under the even one, even under odd
    const w4f upEB = gld<w4f>(bO_up, off_j), upOB = gld<w4f>(bE_up, off_j);      // row y + 2
    w4f ecEA = ownEA, ecOA = ownOA, ecEB = ownEB, ecOB = ownOB;   // epilogue chunks (modes 2-4)
    if (EPI >= 2) {
      const void* ev = (EPI == 4) ? a.epi.dotv : a.epi.other;
      ecEA = epi_chunk(ev, off + rowEA + off_j); ecOA = epi_chunk(ev, off + rowOA + off_j);
      ecEB = epi_chunk(ev, off + rowEB + off_j); ecOB = epi_chunk(ev, off + rowOB + off_j);
    }
    const int gy = a.gy0 + y;
    const int gym = (gy == 0) ? a.gLy - 1 : gy - 1;
    const long gvol = 2 * a.ghalf_vol;
    const char* gc = reinterpret_cast<const char*>(a.gauge);
    const gchar* gxEA = uni(gc + ((long)gy * a.hr) * GB);
    const gchar* gxOA = uni(gc + (a.ghalf_vol + (long)gy * a.hr) * GB);
    const gchar* gyEA = uni(gc + (gvol + (long)gy * a.hr) * GB);
    const gchar* gyOA = uni(gc + (gvol + a.ghalf_vol + (long)gy * a.hr) * GB);
    const gchar* gxEB = uni(gc + ((long)(gy + 1) * a.hr) * GB);
    const gchar* gxOB = uni(gc + (a.ghalf_vol + (long)(gy + 1) * a.hr) * GB);
    const gchar* gyEB = uni(gc + (gvol + (long)(gy + 1) * a.hr) * GB);
    const gchar* gyOB = uni(gc + (gvol + a.ghalf_vol + (long)(gy + 1) * a.hr) * GB);
    const gchar* gyE_dn = uni(gc + (gvol + (long)gym * a.hr) * GB);
    const gchar* gyO_dn = uni(gc + (gvol + a.ghalf_vol + (long)gym * a.hr) * GB);
    const LK uxEA = gld<LK>(gxEA, goff_j), uxOA = gld<LK>(gxOA, goff_j), uyEA = gld<LK>(gyEA, goff_j), uyOA = gld<LK>(gyOA, goff_j);
    const LK uxEB = gld<LK>(gxEB, goff_j), uxOB = gld<LK>(gxOB, goff_j), uyEB = gld<LK>(gyEB, goff_j), uyOB = gld<LK>(gyOB, goff_j);
    const LK ubxA = gld<LK>(sA ? gxEA : gxOA, goff_l), ubxB = gld<LK>(sB ? gxEB : gxOB, goff_l);
    const LK ubyEA = gld<LK>(gyO_dn, goff_j), ubyOA = gld<LK>(gyE_dn, goff_j);
    __builtin_amdgcn_sched_barrier(0);
    char* out = reinterpret_cast<char*>(a.lhs) + off;
    {   // row A, even site: +y = the odd site of row B, back-y link from row y - 1
      const LK bx = sA ? uxOA : ubxA;
      const R lx[4] = {(R)uxEA.x, (R)uyEA.x, (R)bx.x, (R)ubyEA.x}, ly[4] = {(R)uxEA.y, (R)uyEA.y, -(R)bx.y, -(R)ubyEA.y};
      const w4f xr[5] = {sA ? othOA : ownOA, ownOB, sA ? ownOA : othOA, dnEA, ownEA};
      wilson_site<T, 1, EPI>(xr, lx, ly, hw, cw, shE, zE, 0, c0, a, uni(out + rowEA) + off_j, ecEA, live, ed);
    }
    {   // row A, odd site
      const LK bx = sA ? ubxA : uxEA;
      const R lx[4] = {(R)uxOA.x, (R)uyOA.x, (R)bx.x, (R)ubyOA.x}, ly[4] = {(R)uxOA.y, (R)uyOA.y, -(R)bx.y, -(R)ubyOA.y};
      const w4f xr[5] = {sA ? ownEA : othEA, ownEB, sA ? othEA : ownEA, dnOA, ownOA};
      wilson_site<T, 1, EPI>(xr, lx, ly, hw, cw, shO, zO, 1, c0, a, uni(out + rowOA) + off_j, ecOA, live, ed);
    }
    {   // row B, even site: -y = the odd site of row A, back-y link = that site's Uy
      const LK bx = sB ? uxOB : ubxB;
      const R lx[4] = {(R)uxEB.x, (R)uyEB.x, (R)bx.x, (R)uyOA.x}, ly[4] = {(R)uxEB.y, (R)uyEB.y, -(R)bx.y, -(R)uyOA.y};
      const w4f xr[5] = {sB ? othOB : ownOB, upEB, sB ? ownOB : othOB, ownOA, ownEB};
      wilson_site<T, 1, EPI>(xr, lx, ly, hw, cw, shE, zE, 0, c0, a, uni(out + rowEB) + off_j, ecEB, live, ed);
    }
    {   // row B, odd site
      const LK bx = sB ? ubxB : uxEB;
      const R lx[4] = {(R)uxOB.x, (R)uyOB.x, (R)bx.x, (R)uyEA.x}, ly[4] = {(R)uxOB.y, (R)uyOB.y, -(R)bx.y, -(R)uyEA.y};
      const w4f xr[5] = {sB ? ownEB : othEB, upOB, sB ? othEB : ownEB, ownEA, ownOB};
      wilson_site<T, 1, EPI>(xr, lx, ly, hw, cw, shO, zO, 1, c0, a, uni(out + rowOB) + off_j, ecOB, live, ed);
    }
  }
  if (EPI && EPI != 2) wilson_store_partials(a, ed);
}

// one system, overwrite, with the epilogue in mode `m` (1-4; mode 1 needs the own-site chunk: SHAPE 1)
template <typename T, int SHAPE>
static void launch_wilson_epi_s(const WilsonArgs& a, int m, dim3 grid, hipStream_t st) {
  if (m == 1) k_wilson_direct<T, SHAPE, true, false, (SHAPE == 1 ? 1 : 4)><<<grid, BLOCK, 0, st>>>(a);
  else if (m == 2) k_wilson_direct<T, SHAPE, true, false, 2><<<grid, BLOCK, 0, st>>>(a);
  else if (m == 3) k_wilson_direct<T, SHAPE, true, false, 3><<<grid, BLOCK, 0, st>>>(a);
  else k_wilson_direct<T, SHAPE, true, false, 4><<<grid, BLOCK, 0, st>>>(a);
}
template <typename T>
static void launch_wilson_epi(const WilsonArgs& a, int shape, int m, dim3 grid, hipStream_t st) {
  if (shape == 1) launch_wilson_epi_s<T, 1>(a, m, grid, st);
  else if (shape == 3) launch_wilson_epi_s<T, 3>(a, m, grid, st);
  else launch_wilson_epi_s<T, 2>(a, m, grid, st);
}
template <typename T>
static void launch_wilson_pair_epi(const WilsonArgs& a, int m, bool two_rows, dim3 grid, hipStream_t st) {
#define QMG_WP(M) { if (two_rows) k_wilson_pair2<T, true, M><<<grid, BLOCK, 0, st>>>(a); else k_wilson_pair<T, true, false, M><<<grid, BLOCK, 0, st>>>(a); }
  if (m == 1) QMG_WP(1) else if (m == 2) QMG_WP(2) else if (m == 3) QMG_WP(3) else QMG_WP(4)
#undef QMG_WP
}

template <typename T, bool BATCH>
static void launch_wilson_b(const WilsonArgs& a, int shape, bool zero, dim3 grid, hipStream_t st) {
  if (shape == 1) { if (zero) k_wilson_direct<T, 1, true, BATCH><<<grid, BLOCK, 0, st>>>(a); else k_wilson_direct<T, 1, false, BATCH><<<grid, BLOCK, 0, st>>>(a); }
  else if (shape == 3) { if (zero) k_wilson_direct<T, 3, true, BATCH><<<grid, BLOCK, 0, st>>>(a); else k_wilson_direct<T, 3, false, BATCH><<<grid, BLOCK, 0, st>>>(a); }
  else { if (zero) k_wilson_direct<T, 2, true, BATCH><<<grid, BLOCK, 0, st>>>(a); else k_wilson_direct<T, 2, false, BATCH><<<grid, BLOCK, 0, st>>>(a); }
}

}  // namespace qmg

using namespace qmg;

static int wilson_direct_impl(int dtype, const qmg_stencil_desc* d, const void* gauge, int gauge_Ly, int y0, double wilson_coeff, bool scaled, double hop_scale,
                              void* lhs, const void* rhs, const void* halo_lo, const void* halo_hi, unsigned pieces, int nrhs, size_t vec_stride,
                              size_t halo_stride, unsigned mask, int rows, void* stream, const qmg_apply_epilogue* epi = nullptr);

extern "C" {

// The hops of the right-block-Jacobi Wilson stencil from the links: lhs (+)= pieces(H') rhs with H'_mu(x) = H_mu(x) * hop_scale, for
// a cinv that is hop_scale times the identity at every site (real mass, no eo / dof shift: hop_scale = the [0][0] entry qmg_build_rbjacobi
// left in cinv).  `pieces`: all four hops of the processed parities and nothing else (+ the zero bits); otherwise QMG_ERR_UNSUPPORTED.
// In fp64 the result is bit for bit qmg_stencil_apply on the stored right-block-Jacobi hopping through the site kernel.
// Arguments as qmg_wilson_apply_direct.
int qmg_wilson_hops_direct(int dtype, const qmg_stencil_desc* d, const void* gauge, int gauge_Ly, int y0, double wilson_coeff, double hop_scale, void* lhs,
                           const void* rhs, const void* halo_lo, const void* halo_hi, unsigned pieces, int nrhs, size_t vec_stride, size_t halo_stride,
                           unsigned mask, int rows, void* stream) {
  if (pieces & (QMG_P_CLOVER | QMG_P_SHIFT)) return QMG_ERR_UNSUPPORTED;
  return wilson_direct_impl(dtype, d, gauge, gauge_Ly, y0, wilson_coeff, true, hop_scale, lhs, rhs, halo_lo, halo_hi, pieces, nrhs, vec_stride, halo_stride, mask,
                            rows, stream);
}

// lhs (+)= pieces(M_Wilson) rhs from the gauge links; d carries the vectors' lattice (Lx, Ly; nc must be 2) and the shifts,
// its clover / hopping pointers are ignored.  gauge: [2][Lx * gauge_Ly] links in `dtype` on the lattice Lx x gauge_Ly, of
// which the vectors cover rows y0 .. y0 + d->Ly - 1 (the whole lattice: gauge_Ly = d->Ly, y0 = 0).  halo_lo / halo_hi:
// NULL (periodic in y) or the y-slab halos of qmg_halo_exchange; rows as in qmg_stencil_apply_slab.
// Serves: clover + every hop of the processed parities (with or without the shift pieces) and hops only; other piece sets
// return QMG_ERR_UNSUPPORTED (use the stored stencil).  lhs must not alias rhs unless only one parity is written from the other.
int qmg_wilson_apply_direct(int dtype, const qmg_stencil_desc* d, const void* gauge, int gauge_Ly, int y0, double wilson_coeff, void* lhs, const void* rhs,
                            const void* halo_lo, const void* halo_hi, unsigned pieces, int nrhs, size_t vec_stride, size_t halo_stride, unsigned mask, int rows,
                            void* stream) {
  return wilson_direct_impl(dtype, d, gauge, gauge_Ly, y0, wilson_coeff, false, 1.0, lhs, rhs, halo_lo, halo_hi, pieces, nrhs, vec_stride, halo_stride, mask, rows,
                            stream);
}

// The same two entry points with an EPILOGUE on the finished site values (include/qmg_hip.h: qmg_apply_epilogue; qmg_stencil_apply_epi_t is the
// form for the stored stencils): ONE system per launch (exactly one bit of `mask`), overwrite semantics on the processed parities, rows = 0.
int qmg_wilson_apply_direct_epi(int dtype, const qmg_stencil_desc* d, const void* gauge, int gauge_Ly, int y0, double wilson_coeff, void* lhs, const void* rhs,
                                const void* halo_lo, const void* halo_hi, unsigned pieces, int nrhs, size_t vec_stride, size_t halo_stride, unsigned mask,
                                const qmg_apply_epilogue* epi, void* stream) {
  if (!epi) return QMG_ERR_INVALID;
  return wilson_direct_impl(dtype, d, gauge, gauge_Ly, y0, wilson_coeff, false, 1.0, lhs, rhs, halo_lo, halo_hi, pieces, nrhs, vec_stride, halo_stride, mask, 0,
                            stream, epi);
}
int qmg_wilson_hops_direct_epi(int dtype, const qmg_stencil_desc* d, const void* gauge, int gauge_Ly, int y0, double wilson_coeff, double hop_scale, void* lhs,
                               const void* rhs, const void* halo_lo, const void* halo_hi, unsigned pieces, int nrhs, size_t vec_stride, size_t halo_stride,
                               unsigned mask, const qmg_apply_epilogue* epi, void* stream) {
  if (!epi) return QMG_ERR_INVALID;
  if (pieces & (QMG_P_CLOVER | QMG_P_SHIFT)) return QMG_ERR_UNSUPPORTED;
  return wilson_direct_impl(dtype, d, gauge, gauge_Ly, y0, wilson_coeff, true, hop_scale, lhs, rhs, halo_lo, halo_hi, pieces, nrhs, vec_stride, halo_stride, mask,
                            0, stream, epi);
}

}  // extern "C"

static int wilson_direct_impl(int dtype, const qmg_stencil_desc* d, const void* gauge, int gauge_Ly, int y0, double wilson_coeff, bool scaled, double hop_scale,
                              void* lhs, const void* rhs, const void* halo_lo, const void* halo_hi, unsigned pieces, int nrhs, size_t vec_stride,
                              size_t halo_stride, unsigned mask, int rows, void* stream, const qmg_apply_epilogue* epi) {
  if (!valid_dtype(dtype) || !d || !gauge || !lhs || !rhs || nrhs < 1 || nrhs > 16 || rows < 0 || rows > 2) return QMG_ERR_INVALID;
  if (!valid_lattice(d->Lx, d->Ly) || !valid_lattice(d->Lx, gauge_Ly) || y0 < 0 || (y0 & 1) || y0 + d->Ly > gauge_Ly) return QMG_ERR_INVALID;
  if (d->nc != 2) return QMG_ERR_UNSUPPORTED;
  if ((halo_lo == nullptr) != (halo_hi == nullptr)) return QMG_ERR_INVALID;
  if (!halo_lo && (gauge_Ly != d->Ly || rows != 0)) return QMG_ERR_INVALID;   // a slab needs its halos
  if (nrhs > 1 && (vec_stride < (size_t)d->Lx * d->Ly * 2 || (halo_lo && halo_stride < (size_t)d->Lx * 2))) return QMG_ERR_INVALID;
  WilsonArgs a;
  a.gauge = gauge; a.lhs = lhs; a.rhs = rhs;
  a.hr = d->Lx / 2; a.Ly = d->Ly; a.gLy = gauge_Ly; a.gy0 = y0;
  a.half_vol = (long)a.hr * d->Ly; a.ghalf_vol = (long)a.hr * gauge_Ly;
  a.pieces = pieces; a.vec_stride = (long)vec_stride; a.w = wilson_coeff; a.hop_scale = hop_scale;
  a.nrhs = 0;
  for (int k = 0; k < 16; k++) a.ridx[k] = 0;
  for (int k = 0; k < nrhs; k++)
    if ((mask >> k) & 1u) a.ridx[a.nrhs++] = k;
  if (a.nrhs == 0) return QMG_SUCCESS;
  a.epi = no_epilogue();
  if (epi) {
    if (a.nrhs != 1) return QMG_ERR_UNSUPPORTED;      // one system per launch
    if (epi->other == lhs || epi->dotv == lhs) return QMG_ERR_INVALID;
    a.epi.on = 1;
    a.epi.other = epi->other; a.epi.other_scale = epi->other_scale; a.epi.acc_scale = epi->acc_scale; a.epi.dotv = epi->dotv;
    if (epi->other && epi->dotv && epi->other != epi->dotv) return QMG_ERR_UNSUPPORTED;   // two different extra vectors: the separate passes
  }
  for (int i = 0; i < 2; i++) { a.shift[i] = d->shift[i]; a.eo_shift[i] = d->eo_shift[i]; a.dof_shift[i] = d->dof_shift[i]; }
  const unsigned even_bits = QMG_P_CLOVER_E | QMG_P_EO | QMG_P_SHIFT_E | QMG_P_ZERO_E;
  const unsigned odd_bits = QMG_P_CLOVER_O | QMG_P_OE | QMG_P_SHIFT_O | QMG_P_ZERO_O;
  const bool ev = pieces & even_bits, od = pieces & odd_bits;
  if (!ev && !od) return QMG_SUCCESS;
  a.par_first = ev ? 0 : 1;
  a.par_count = (ev && od) ? 2 : 1;
  a.halo_lo = halo_lo; a.halo_hi = halo_hi; a.halo_stride = (long)halo_stride;
  a.boundary_only = rows == 2;
  a.y_first = rows == 1 ? 1 : 0;
  a.y_count = rows == 2 ? 2 : rows == 1 ? d->Ly - 2 : d->Ly;
  if (a.y_count <= 0) return QMG_SUCCESS;
  a.nrows = a.y_count * a.par_count;
  int sh[2] = {0, 0};
  bool zero = true;
  for (int q = 0; q < a.par_count; q++) {
    const int p = (a.par_count == 2) ? q : a.par_first;
    const bool cl = (pieces >> p) & 1u, shf = (pieces >> (10 + p)) & 1u;
    const unsigned hm = (pieces >> (2 + 4 * p)) & 0xFu;
    sh[q] = (hm == 0xFu) ? (cl ? 1 : (shf ? 0 : 2)) : 0;
    if (!((pieces >> (12 + p)) & 1u)) zero = false;
  }
  int shape = (a.par_count == 2 && sh[0] != sh[1]) ? 0 : sh[0];
  if (shape == 0 || (scaled && shape != 2)) return QMG_ERR_UNSUPPORTED;
  if (scaled) shape = 3;
  if (lhs == rhs && (shape == 1 || a.par_count == 2)) return QMG_ERR_INVALID;
  if (epi && (!zero || lhs == rhs)) return QMG_ERR_INVALID;   // an epilogue needs overwrite semantics and a separate output
  const int lps = dtype == QMG_C64 ? 2 : 1;
  const long lanes = (long)a.hr * lps;
  dim3 grid((unsigned)((lanes + BLOCK - 1) / BLOCK), a.nrows > 65535 ? 65535u : (unsigned)a.nrows);
  hipStream_t st = as_stream(stream);
  // the epilogue's dot partials: one per wavefront of the launch that is chosen below
  long epi_npart = 0;
  // epilogue mode (wilson_site): 1 dots against the right-hand side (own-site chunk; needs SHAPE 1, else the vector is loaded: mode 4),
  // 2 combination only, 3 combination + dots against the same vector, 4 dots against a separate vector
  int emode = 0;
  if (a.epi.on) {
    if (!a.epi.other && !a.epi.dotv) emode = 2;                       // a pure scaling of the result (other_scale is moot): served by mode 2 with no vector
    else if (a.epi.other && !a.epi.dotv) emode = 2;
    else if (a.epi.other) emode = 3;
    else emode = (a.epi.dotv == rhs && shape == 1) ? 1 : 4;
  }
  // launches with the MR dots keep the number of partials (one per wavefront) small enough for the one-block second stage: the blocks
  // walk several rows each (every kernel W form has the row loop; capping grid.y costs nothing)
  auto epi_cap = [&](dim3& g) { if (a.epi.on && a.epi.dotv) { const unsigned cap = g.x >= 2048u ? 1u : 2048u / g.x; if (g.y > cap) g.y = cap; } };
  auto epi_begin = [&](dim3 g) -> bool {
    if (!(a.epi.on && a.epi.dotv)) return true;
    epi_npart = (long)g.x * (long)g.y * (BLOCK / WAVE);
    a.epi.part = mr_epilogue_begin(1, epi_npart);
    a.epi.npart = epi_npart;
    return a.epi.part != nullptr;
  };
  auto epi_finish = [&]() -> int {
    if (!epi_npart) return QMG_SUCCESS;
    const unsigned char id0 = (unsigned char)a.ridx[0];
    return mr_epilogue_finish(&id0, 1, epi_npart, st);
  };
  // (an epilogue that loads a vector -- modes 2-4 -- would take the two-row form to 170-178 VGPRs, two wavefronts per SIMD: those go through the one-row form)
  if (shape == 1 && a.par_count == 2 && g_wilson_pair >= 2 && a.nrhs == 1 && !a.boundary_only && a.y_count % 2 == 0 && emode <= 1) {
    // one system, an even run of consecutive rows: kernel W2 on two rows per lane group
    const int ny = a.y_count / 2;   // (capping grid.y -- a row-pair loop per block -- changes nothing: 0.29-0.31 ms at every cap)
    dim3 gridp(grid.x, ny > 65535 ? 65535u : (unsigned)ny);
    epi_cap(gridp);
    if (!epi_begin(gridp)) return QMG_ERR_HIP;
    if (a.epi.on) { if (dtype == QMG_C64) launch_wilson_pair_epi<double>(a, emode, true, gridp, st); else launch_wilson_pair_epi<float>(a, emode, true, gridp, st); }
    else if (dtype == QMG_C64) { if (zero) k_wilson_pair2<double, true><<<gridp, BLOCK, 0, st>>>(a); else k_wilson_pair2<double, false><<<gridp, BLOCK, 0, st>>>(a); }
    else { if (zero) k_wilson_pair2<float, true><<<gridp, BLOCK, 0, st>>>(a); else k_wilson_pair2<float, false><<<gridp, BLOCK, 0, st>>>(a); }
    QMG_LAUNCH_CHECK();
    return epi_finish();
  }
  if (shape == 1 && a.par_count == 2 && g_wilson_pair) {   // the full operator: both parities of a column per lane group (kernel W2)
    dim3 gridp(grid.x, a.y_count > 65535 ? 65535u : (unsigned)a.y_count);
    epi_cap(gridp);
    if (!epi_begin(gridp)) return QMG_ERR_HIP;
    if (a.epi.on) { if (dtype == QMG_C64) launch_wilson_pair_epi<double>(a, emode, false, gridp, st); else launch_wilson_pair_epi<float>(a, emode, false, gridp, st); }
    else if (dtype == QMG_C64) {
      if (a.nrhs == 1) { if (zero) k_wilson_pair<double, true, false><<<gridp, BLOCK, 0, st>>>(a); else k_wilson_pair<double, false, false><<<gridp, BLOCK, 0, st>>>(a); }
      else { if (zero) k_wilson_pair<double, true, true><<<gridp, BLOCK, 0, st>>>(a); else k_wilson_pair<double, false, true><<<gridp, BLOCK, 0, st>>>(a); }
    } else {
      if (a.nrhs == 1) { if (zero) k_wilson_pair<float, true, false><<<gridp, BLOCK, 0, st>>>(a); else k_wilson_pair<float, false, false><<<gridp, BLOCK, 0, st>>>(a); }
      else { if (zero) k_wilson_pair<float, true, true><<<gridp, BLOCK, 0, st>>>(a); else k_wilson_pair<float, false, true><<<gridp, BLOCK, 0, st>>>(a); }
    }
    QMG_LAUNCH_CHECK();
    return epi_finish();
  }
  epi_cap(grid);
  if (!epi_begin(grid)) return QMG_ERR_HIP;
  if (a.epi.on) { if (dtype == QMG_C64) launch_wilson_epi<double>(a, shape, emode, grid, st); else launch_wilson_epi<float>(a, shape, emode, grid, st); }
  else if (dtype == QMG_C64) { if (a.nrhs == 1) launch_wilson_b<double, false>(a, shape, zero, grid, st); else launch_wilson_b<double, true>(a, shape, zero, grid, st); }
  else { if (a.nrhs == 1) launch_wilson_b<float, false>(a, shape, zero, grid, st); else launch_wilson_b<float, true>(a, shape, zero, grid, st); }
  QMG_LAUNCH_CHECK();
  return epi_finish();
}

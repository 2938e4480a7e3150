// qmg_transfer.hip -- restrict / prolong (transfer/transfer.h:455-511), block
// orthonormalisation (:514-607) and the Galerkin coarse-operator build (operators/coarse.h:90-444).
//
// The reference walks an explicit one-to-many `coarse_map[i][j]` (transfer.h:410-448) and gathers
// from nvec separate vector-major arrays.  On the GPU the map is never materialised: the blocks
// are regular rectangles, so the coarse site of a fine element follows from its (parity,y,j)
// position, and for an even block width bx the (bx/2)*nc_f elements a block owns on one fine
// half-row are CONTIGUOUS (x = 2j+s, s in {0,1}, never crosses a block edge).  Both kernels
// therefore stream the null vectors in their native vector-major layout with coalesced loads:
// the traffic is nvec * size_cv_f * 16 B, the algorithmic minimum of SURVEY 8d.
#include <string.h>

#include "qmg_common.h"

namespace qmg {

struct XferGeom {
  int fhr, fLy, fnc;     // fine half-row length, rows, dof
  int chr, cLy, cnc;     // coarse
  int bx, by;            // block extents (fine sites)
  long fhalf_vol, chalf_vol;
  long fsize;            // fine size_cv (null-vector stride)
};

__device__ __forceinline__ long coarse_site_index(const XferGeom& g, int cx, int cy) {
  // lattice.h:75-81 on the coarse lattice
  const int p = (cx + cy) & 1;
  return (long)(cy + p * g.cLy) * g.chr + (cx >> 1);
}

// ---------------- prolong: fine[k] += sum_d null[d][k] * coarse[ci(k)*cnc + d] ----------------
// T = storage scalar of the null vectors and of both vectors; arithmetic in fp64 registers.
// W = elements per lane: 1, or 2 for complex<float> with an even fnc -- then every access is one 16-byte load / store as in
// fp64 (with 8-byte accesses the same number of instructions moves half the bytes: the fp32 prolong ran at 0.69 of peak).
// NS = storage scalar of the null vectors when it differs from the vectors' (complex<float> null vectors under complex<double> vectors: a
// preconditioner level's narrow copy -- W = 2 there, so that a lane's null-vector load is still 16 bytes).
template <typename T, int W, typename NS = T>
__global__ __launch_bounds__(BLOCK) void k_prolong(const void* __restrict__ nullv, int nvec, const void* __restrict__ coarse,
                                                   void* __restrict__ fine, const XferGeom g) {
  const long row_packs = (long)g.fhr * g.fnc / W;
  const int nrows = 2 * g.fLy;
  for (int row = blockIdx.y; row < nrows; row += gridDim.y) {
    const int p = row / g.fLy, y = row - p * g.fLy;
    const int s = (y + p) & 1;
    const int cy = y / g.by;
    for (long t = (long)blockIdx.x * BLOCK + threadIdx.x; t < row_packs; t += (long)gridDim.x * BLOCK) {
      const int j = (int)(t * W / g.fnc);                       // the W elements of a pack belong to one site (fnc even)
      const int cx = (2 * j + s) / g.bx;
      const long ci = coarse_site_index(g, cx, cy);
      const long kp = ((long)p * g.fhalf_vol + (long)y * g.fhr) * g.fnc / W + t;   // pack index in the fine vector
      const long cv = ci * g.cnc;
      cplx acc[W];
      ldc_pack<T, W>(fine, kp, acc);
      // the null vectors are read exactly once: non-temporal, 8 loads in flight per lane before the FMAs
      int d = 0;
      for (; d + 8 <= nvec; d += 8) {
        cplx v[8][W];
#pragma unroll
        for (int q = 0; q < 8; q++) ldc_pack_nt<NS, W>(nullv, (long)(d + q) * (g.fsize / W) + kp, v[q]);
#pragma unroll
        for (int q = 0; q < 8; q++) {
          const cplx c = ldc<T>(coarse, cv + d + q);
#pragma unroll
          for (int w = 0; w < W; w++) cmac(acc[w], v[q][w], c);
        }
      }
      for (; d < nvec; d++) {
        cplx v[W];
        ldc_pack<NS, W>(nullv, (long)d * (g.fsize / W) + kp, v);
        const cplx c = ldc<T>(coarse, cv + d);
#pragma unroll
        for (int w = 0; w < W; w++) cmac(acc[w], v[w], c);
      }
      stc_pack<T, W>(fine, kp, acc);
    }
  }
}

// ---------------- restrict: coarse[ci*cnc + d] += sum_{k in block ci} conj(null[d][k]) fine[k] ----------------
// Block = NG consecutive coarse sites of one coarse row; TPG threads per coarse site walk the
// G = (bx/2)*fnc contiguous elements it owns on each of its 2*by fine half-rows, accumulating DC
// null vectors at a time in registers; LDS sums the TPG partials.  One writer per (site, d): no atomics.
constexpr int XFER_DC = 8;

template <typename T, int W, typename NS = T>
__global__ __launch_bounds__(BLOCK) void k_restrict(const void* __restrict__ nullv, int nvec, const void* __restrict__ fine,
                                                    void* __restrict__ coarse, const XferGeom g, int NG, int TPG) {
  extern __shared__ __align__(16) unsigned char smem_raw[];
  cplx* red = reinterpret_cast<cplx*>(smem_raw);   // [BLOCK][XFER_DC]
  const int G = (g.bx / 2) * g.fnc / W;          // packs of W elements (16 bytes) a coarse site owns on a fine half-row
  const int grp = threadIdx.x / TPG;
  const int l = threadIdx.x - grp * TPG;
  const int cLx = 2 * g.chr;
  for (int cy = blockIdx.y; cy < g.cLy; cy += gridDim.y) {
    const int cx = blockIdx.x * NG + grp;
    const bool active = grp < NG && cx < cLx;
    for (int d0 = 0; d0 < nvec; d0 += XFER_DC) {
      const int dn = (nvec - d0 < XFER_DC) ? nvec - d0 : XFER_DC;
      cplx acc[XFER_DC];
#pragma unroll
      for (int q = 0; q < XFER_DC; q++) acc[q] = cmake(0.0, 0.0);
      if (active) {
        for (int rr = 0; rr < 2 * g.by; rr++) {
          const int p = rr / g.by;
          const int y = cy * g.by + (rr - p * g.by);
          const long base = ((long)p * g.fhalf_vol + (long)y * g.fhr + (long)cx * (g.bx / 2)) * g.fnc / W;
          for (int el = l; el < G; el += TPG) {
            const long kp = base + el;
            // every load of the element first, in the storage form (qmg_common.h: a widening or a branch per load serialises them)
            cplx f[W];
            typename RawP<float, 2>::type fr32;   // (complex<float> vectors: kept raw until the null vectors are requested)
            if constexpr (sizeof(T) == 4 && W == 2) fr32 = ld_rawp<T, W>(fine, kp);
            else ldc_pack<T, W>(fine, kp, f);     // (complex<double>: nothing to widen; complex<float> one element per lane: the 8-byte form)
            typename RawP<NS, W>::type vr[XFER_DC];
#pragma unroll
            for (int q = 0; q < XFER_DC; q++) {
              vr[q] = zero_rawp<NS, W>();
              if (q < dn) vr[q] = ld_rawp_nt<NS, W>(nullv, (long)(d0 + q) * (g.fsize / W) + kp);   // read-once stream: non-temporal
            }
            if constexpr (sizeof(T) == 4 && W == 2) widen_rawp<T, W>(fr32, f);
#pragma unroll
            for (int q = 0; q < XFER_DC; q++)
              if (q < dn) {
                cplx v[W];
                widen_rawp<NS, W>(vr[q], v);
#pragma unroll
                for (int w = 0; w < W; w++) cmac_conj(acc[q], v[w], f[w]);
              }
          }
        }
      }
      __syncthreads();
#pragma unroll
      for (int q = 0; q < XFER_DC; q++) red[threadIdx.x * XFER_DC + q] = acc[q];
      __syncthreads();
      // thread (grp2, q) sums the TPG partials of its coarse site
      for (int w = threadIdx.x; w < NG * dn; w += BLOCK) {
        const int g2 = w / dn, q = w - g2 * dn;
        const int cx2 = blockIdx.x * NG + g2;
        if (cx2 >= cLx) continue;
        cplx t = cmake(0.0, 0.0);
        for (int u = 0; u < TPG; u++) t = cadd(t, red[(g2 * TPG + u) * XFER_DC + q]);
        const long o = coarse_site_index(g, cx2, cy) * g.cnc + d0 + q;
        stc<T>(coarse, o, cadd(ldc<T>(coarse, o), t));
      }
    }
    __syncthreads();
  }
}

// Generic fallback (odd bx, bx = 1, ...): one thread per (coarse site, d), walking the block by coordinates.
template <typename T>
__global__ __launch_bounds__(BLOCK) void k_restrict_generic(const void* __restrict__ nullv, int nvec, const void* __restrict__ fine,
                                                            void* __restrict__ coarse, const XferGeom g) {
  const int cLx = 2 * g.chr;
  const long total = (long)cLx * g.cLy * nvec;
  for (long t = (long)blockIdx.x * BLOCK + threadIdx.x; t < total; t += (long)gridDim.x * BLOCK) {
    const int d = (int)(t % nvec);
    const long cs = t / nvec;
    const int cx = (int)(cs % cLx), cy = (int)(cs / cLx);
    cplx acc = cmake(0.0, 0.0);
    for (int yy = 0; yy < g.by; yy++)
      for (int xx = 0; xx < g.bx; xx++) {
        const int x = cx * g.bx + xx, y = cy * g.by + yy;
        const int p = (x + y) & 1;
        const long site = (long)(y + p * g.fLy) * g.fhr + (x >> 1);
        for (int c = 0; c < g.fnc; c++) {
          const long k = site * g.fnc + c;
          cmac_conj(acc, ldc<T>(nullv, (long)d * g.fsize + k), ldc<T>(fine, k));
        }
      }
    const long o = coarse_site_index(g, cx, cy) * g.cnc + d;
    stc<T>(coarse, o, cadd(ldc<T>(coarse, o), acc));
  }
}

// The (up to 8) systems of one pass, indexed by compile-time slot numbers only: a by-value struct indexed at run time
// (BatchIdx::id[s0 + q]) is copied to scratch memory by the compiler.  Slots beyond `n` alias slot 0 (computed, discarded).
struct PassIds { int n; int id[8]; };
// run-time slot -> system id without indexing the struct (a chain of selects on registers)
__device__ __forceinline__ int pick_id(const PassIds& p, int q) {
  const int a = (q & 1) ? p.id[1] : p.id[0], b = (q & 1) ? p.id[3] : p.id[2], c = (q & 1) ? p.id[5] : p.id[4], d = (q & 1) ? p.id[7] : p.id[6];
  const int ab = (q & 2) ? b : a, cd = (q & 2) ? d : c;
  return (q & 4) ? cd : ab;
}
static PassIds make_pass(const BatchIdx& bi, int s0) {
  PassIds p;
  p.n = (bi.n - s0 < 8) ? bi.n - s0 : 8;
  for (int q = 0; q < 8; q++) p.id[q] = bi.id[s0 + (q < p.n ? q : 0)];
  return p;
}

// ---------------------------------------------------------------------------------------------------------------------
// Transfer for a lock-step batch (2..16 systems): the null vectors -- nvec of the (nvec + 2k) size_cv_f complex a call
// moves, and the same for every system -- are streamed ONCE for up to KB systems.  Both kernels work on the fine elements
// that consecutive coarse sites of one coarse row own: 2*by fine half-rows (both parities) x G = (bx/2)*fnc contiguous
// elements per site, so every wavefront access is whole 128-byte lines.
//
// Prolong: the coarse values of a tile of SX sites (SX x nvec x KB complex, a few KB) are staged in LDS once; a thread
// owns a fine element, streams its nvec null-vector entries non-temporally (4 in flight) and reads the KB coarse values
// of each d as LDS broadcasts (all lanes of a site read the same address; the site stride is padded so that different
// sites of a row hit different banks).  (The first version looped 8 x nvec coarse values per element through L1:
// address-bound at 2 TB/s, profiles/r01_kernel_rooflines.json.)
template <typename T, int KB, int NVB = 4>
__global__ __launch_bounds__(BLOCK) void k_bprolong_tile(const void* __restrict__ nullv, int nvec, const void* __restrict__ coarse, void* __restrict__ fine,
                                                         const XferGeom g, const PassIds ids, long cstride, long fstride, int SX) {
  extern __shared__ __align__(16) unsigned char smem_raw[];
  // the coarse tile in the vectors' STORAGE type: complex<float> systems keep 8-byte entries (the 8-system form reads nvec x 8 of them per fine
  // element: at 16 bytes each the LDS reads were a third of the kernel's time), widened when they are multiplied
  typedef typename CStore<T>::type lt;
  lt* cl = reinterpret_cast<lt*>(smem_raw);          // [SX][nvec * KB + 1]
  const int ns = (ids.n < KB) ? ids.n : KB;
  const int G = (g.bx / 2) * g.fnc, R = 2 * g.by;
  const int cLx = 2 * g.chr;
  const int cx0 = blockIdx.x * SX;
  const int nsx = (cLx - cx0 < SX) ? cLx - cx0 : SX;
  const int sstride = nvec * KB + 1;
  long fo[KB];
#pragma unroll
  for (int q = 0; q < KB; q++) fo[q] = (long)ids.id[q] * fstride;   // unused slots alias slot 0 (computed, discarded)
  for (int cy = blockIdx.y; cy < g.cLy; cy += gridDim.y) {
    __syncthreads();   // the previous tile's reads are done
    for (int t = threadIdx.x; t < nsx * KB * nvec; t += BLOCK) {   // d fastest: coalesced runs of nvec coarse values
      const int d = t % nvec, q = (t / nvec) % KB, s = t / (nvec * KB);
      const long ci = coarse_site_index(g, cx0 + s, cy);
      lt cv; cv.x = 0; cv.y = 0;
      if (q < ns) cv = reinterpret_cast<const lt*>(coarse)[(long)pick_id(ids, q) * cstride + ci * g.cnc + d];
      cl[s * sstride + d * KB + q] = cv;
    }
    __syncthreads();
    const int row_w = nsx * G;
    for (int t = threadIdx.x; t < R * row_w; t += BLOCK) {
      const int rr = t / row_w, u = t - rr * row_w, s = u / G;
      const int p = rr / g.by, y = cy * g.by + (rr - p * g.by);
      const long e = ((long)p * g.fhalf_vol + (long)y * g.fhr + (long)cx0 * (g.bx / 2)) * g.fnc + u;
      const lt* cs = cl + s * sstride;
      auto wide = [](const lt v) { return cmake((double)v.x, (double)v.y); };
      // the fine values this element is added to: requested now, in the storage form, widened and added after the products (an accumulator that
      // STARTS from the widened load makes the first multiply-add wait for it, and the widenings behind the loads made them wait for one another)
      typename RawC<T>::type fraw[KB];
#pragma unroll
      for (int q = 0; q < KB; q++) fraw[q] = ld_raw<T>(fine, fo[q] + e);
      cplx acc[KB];
#pragma unroll
      for (int q = 0; q < KB; q++) acc[q] = cmake(0.0, 0.0);
      int d = 0;
      // NVB null-vector entries requested together and kept in their STORAGE type until they are used: a thread has NVB x 8 (complex<float>) or
      // NVB x 16 bytes in flight; with 4 of them the complex<float> form had 32 B per thread outstanding and ran at 0.42-0.48 of the HBM rate
      // (NVB = 12 is instantiated for complex<float> with >= 12 null vectors: 2048^2 -> 512^2 x 24, 8 systems 0.90 -> 0.82 ms, 512^2 -> 128^2 0.58 -> 0.47;
      // compiled into the nvec = 8 launches as well it cost them a wavefront of occupancy: 1.88 -> 2.14 ms)
      if constexpr (sizeof(T) == 4 && NVB > 4) {
        for (; d + NVB <= nvec; d += NVB) {
          long long nv[NVB];   // raw bits of a complex<float>
#pragma unroll
          for (int w = 0; w < NVB; w++) nv[w] = __builtin_nontemporal_load(reinterpret_cast<const long long*>(nullv) + (long)(d + w) * g.fsize + e);
#pragma unroll
          for (int w = 0; w < NVB; w++) {
            const cplx nw = cmake((double)__int_as_float((int)(nv[w] & 0xFFFFFFFFll)), (double)__int_as_float((int)(nv[w] >> 32)));
#pragma unroll
            for (int q = 0; q < KB; q++) cmac(acc[q], nw, wide(cs[(d + w) * KB + q]));
          }
        }
      }
      for (; d + 4 <= nvec; d += 4) {
        cplx nv[4];
#pragma unroll
        for (int w = 0; w < 4; w++) nv[w] = ldc_nt<T>(nullv, (long)(d + w) * g.fsize + e);
#pragma unroll
        for (int w = 0; w < 4; w++)
#pragma unroll
          for (int q = 0; q < KB; q++) cmac(acc[q], nv[w], wide(cs[(d + w) * KB + q]));
      }
      for (; d < nvec; d++) {
        const cplx nv = ldc_nt<T>(nullv, (long)d * g.fsize + e);
#pragma unroll
        for (int q = 0; q < KB; q++) cmac(acc[q], nv, wide(cs[d * KB + q]));
      }
#pragma unroll
      for (int q = 0; q < KB; q++)
        if (q < ns) stc<T>(fine, fo[q] + e, cadd(widen_rawc<T>(fraw[q]), acc[q]));
    }
  }
}

// Restrict: a half wavefront (32 lanes) owns one coarse site, a workgroup 8 consecutive sites of a coarse row.  Lane l
// walks elements l, l+32, ... of the site's block (row-major over its 2*by runs of G elements), keeps the KB fine values
// of an element in registers and accumulates DC = 32/KB null vectors x KB systems of partial sums (64 doubles).  The sum
// over the 32 lanes is a RECURSIVE HALVING: at step m in {1,2,4,8,16} a lane keeps the half of its values selected by
// bit m of its id and adds the partner's (lane ^ m) copy of that half -- 32+16+8+4+2 = 62 exchanges instead of the
// 64 x 5 of a plain butterfly (which made the first version shuffle-bound) -- and ends with ONE complex sum, the one
// for (d, system) = bit-reversed lane id: one writer per output, fixed order, deterministic.
template <typename T, int KB>
__global__ __launch_bounds__(BLOCK) void k_brestrict_tile(const void* __restrict__ nullv, int nvec, const void* __restrict__ fine, void* __restrict__ coarse,
                                                          const XferGeom g, const PassIds ids, long cstride, long fstride) {
  constexpr int DC = 32 / KB;        // null vectors per pass
  constexpr int NV = 64;             // doubles per lane per pass: DC * KB complex
  const int ns = (ids.n < KB) ? ids.n : KB;
  const int cLx = 2 * g.chr;
  const long ncs = (long)cLx * g.cLy;
  const int G = (g.bx / 2) * g.fnc;
  const int nel = 2 * g.by * G;
  const int grp = threadIdx.x >> 5, l = threadIdx.x & 31;
  long fo[KB];
#pragma unroll
  for (int q = 0; q < KB; q++) fo[q] = (long)ids.id[q] * fstride;
  // the (d, system) pair this lane ends up holding: bit-reversed lane id
  const int pair = ((l & 1) << 4) | ((l & 2) << 2) | (l & 4) | ((l & 8) >> 2) | ((l & 16) >> 4);
  const int my_dq = pair / KB, my_q = pair - my_dq * KB;
  for (long cs = (long)blockIdx.x * (BLOCK / 32) + grp; cs < ncs; cs += (long)gridDim.x * (BLOCK / 32)) {
    const int cy = (int)(cs / cLx), cx = (int)(cs - (long)cy * cLx);
    const long ci = coarse_site_index(g, cx, cy);
    for (int d0 = 0; d0 < nvec; d0 += DC) {
      const int dn = (nvec - d0 < DC) ? nvec - d0 : DC;
      double v[NV];
#pragma unroll
      for (int i = 0; i < NV; i++) v[i] = 0.0;
      for (int t = l; t < nel; t += 32) {
        const int rr = t / G, el = t - rr * G;
        const int p = rr / g.by, y = cy * g.by + (rr - p * g.by);
        const long e = ((long)p * g.fhalf_vol + (long)y * g.fhr + (long)cx * (g.bx / 2)) * g.fnc + el;
        typename RawC<T>::type fraw[KB], nraw[DC];   // every load of the element first, in the storage form
#pragma unroll
        for (int q = 0; q < KB; q++) fraw[q] = ld_raw<T>(fine, fo[q] + e);
#pragma unroll
        for (int dq = 0; dq < DC; dq++) {
          nraw[dq] = zero_rawc<T>();
          if (dq < dn) nraw[dq] = ld_raw_nt<T>(nullv, (long)(d0 + dq) * g.fsize + e);
        }
        cplx f[KB];
#pragma unroll
        for (int q = 0; q < KB; q++) f[q] = widen_rawc<T>(fraw[q]);
#pragma unroll
        for (int dq = 0; dq < DC; dq++) {
          if (dq < dn) {
            const cplx nv = widen_rawc<T>(nraw[dq]);
#pragma unroll
            for (int q = 0; q < KB; q++) {   // += conj(nv) f
              double& ar = v[(dq * KB + q) * 2];
              double& ai = v[(dq * KB + q) * 2 + 1];
              ar = fma(nv.x, f[q].x, ar); ar = fma(nv.y, f[q].y, ar);
              ai = fma(nv.x, f[q].y, ai); ai = fma(-nv.y, f[q].x, ai);
            }
          }
        }
      }
      // recursive halving over the 32 lanes of the group (all index arithmetic is compile-time after unrolling)
#pragma unroll
      for (int step = 0; step < 5; step++) {
        const int m = 1 << step;
        const int half = NV >> (step + 1);
        const bool up = (l & m) != 0;
#pragma unroll
        for (int i = 0; i < half; i++) {
          const double lo = v[i], hi = v[i + half];
          const double send = up ? lo : hi;
          const double keep = up ? hi : lo;
          double recv;
          if (m == 1) recv = lane_xor1(send);
          else if (m == 2) recv = lane_xor2(send);
          else recv = __shfl_xor(send, m);
          v[i] = keep + recv;
        }
      }
      if (my_dq < dn && my_q < ns) {
        const long o = (long)pick_id(ids, my_q) * cstride + ci * g.cnc + d0 + my_dq;
        const cplx c = ldc<T>(coarse, o);
        stc<T>(coarse, o, cmake(c.x + v[0], c.y + v[1]));
      }
    }
  }
}

// ---- small helper kernels for block-ortho and the coarse build ----
// v[i*stride] = 1/sqrt(re v[i*stride]) for every element (inv_real_sqrt over the whole coarse vector, transfer.h:583)
__global__ __launch_bounds__(BLOCK) void k_inv_real_sqrt(cplx* __restrict__ v, long n) {
  for (long i = (long)blockIdx.x * BLOCK + threadIdx.x; i < n; i += (long)gridDim.x * BLOCK) v[i] = cmake(1.0 / sqrt(v[i].x), 0.0);
}
// bi-ortho helpers (transfer.h:366-380, 707, 736): MODE 0: z -> polar(1/sqrt|z|, arg z) ; 1: z -> |z| ; 2: z -> conj z.
// (MODE is a template parameter on purpose: with a run-time three-way branch hipcc 7.2 -O3 emitted code whose
//  mode-2 path stored a zero imaginary part -- seen in the ISA and caught by the parity test.)
template <int MODE>
__global__ __launch_bounds__(BLOCK) void k_elementwise(cplx* __restrict__ v, long n) {
  for (long i = (long)blockIdx.x * BLOCK + threadIdx.x; i < n; i += (long)gridDim.x * BLOCK) {
    const cplx z = v[i];
    if (MODE == 2) { v[i] = cmake(z.x, -z.y); continue; }
    const double a = sqrt(z.x * z.x + z.y * z.y);
    if (MODE == 0) { const double s = 1.0 / (a * sqrt(a)); v[i] = cmake(z.x * s, z.y * s); }
    else v[i] = cmake(a, 0.0);
  }
}
// chol[site*nc2 + off] = src[site*cnc] (or its inverse)   (copy_vector_blas at transfer.h:560,592)
__global__ __launch_bounds__(BLOCK) void k_chol_store(cplx* __restrict__ chol, const cplx* __restrict__ src, long cvol, int cnc, int off, int invert) {
  for (long s = (long)blockIdx.x * BLOCK + threadIdx.x; s < cvol; s += (long)gridDim.x * BLOCK) {
    cplx v = src[s * cnc];
    if (invert) { const double m = v.x * v.x + v.y * v.y; v = cmake(v.x / m, -v.y / m); }
    chol[s * (long)cnc * cnc + off] = v;
  }
}
// unit probe: tc[i*cnc + color] = 1 for coarse sites lo <= i < hi, zero elsewhere
__global__ __launch_bounds__(BLOCK) void k_unit_probe(cplx* __restrict__ tc, long cvol, int cnc, int color, long lo, long hi) {
  const long n = cvol * cnc;
  for (long t = (long)blockIdx.x * BLOCK + threadIdx.x; t < n; t += (long)gridDim.x * BLOCK) {
    const long i = t / cnc;
    const int c = (int)(t - i * cnc);
    tc[t] = cmake((c == color && i >= lo && i < hi) ? 1.0 : 0.0, 0.0);
  }
}
// scatter a probe result into column `color` of the coarse clover / hopping (coarse.h:169-171,222-233)
__global__ __launch_bounds__(BLOCK) void k_probe_scatter(cplx* __restrict__ clover, cplx* __restrict__ hop_dir, const cplx* __restrict__ tc,
                                                         long cvol, int cnc, int color, long lo, long hi, int fold) {
  const long n = cvol * cnc;
  for (long t = (long)blockIdx.x * BLOCK + threadIdx.x; t < n; t += (long)gridDim.x * BLOCK) {
    const long i = t / cnc;
    const int c = (int)(t - i * cnc);
    const bool same = (i >= lo && i < hi);
    const long o = (i * cnc + c) * cnc + color;
    if (same || fold || !hop_dir) clover[o] = cadd(clover[o], tc[t]);
    else hop_dir[o] = cadd(hop_dir[o], tc[t]);
  }
}

static int make_geom(XferGeom* g, int fLx, int fLy, int fnc, int cLx, int cLy, int cnc) {
  if (!valid_lattice(fLx, fLy) || !valid_lattice(cLx, cLy) || fnc < 1 || cnc < 1) return QMG_ERR_INVALID;
  if (fLx % cLx || fLy % cLy) return QMG_ERR_INVALID;   // transfer.h:130-134
  g->fhr = fLx / 2; g->fLy = fLy; g->fnc = fnc;
  g->chr = cLx / 2; g->cLy = cLy; g->cnc = cnc;
  g->bx = fLx / cLx; g->by = fLy / cLy;
  g->fhalf_vol = (long)g->fhr * fLy;
  g->chalf_vol = (long)g->chr * cLy;
  g->fsize = 2 * g->fhalf_vol * fnc;
  return QMG_SUCCESS;
}

// Restrict, blocks of at most 32 elements (the fine level: 4x4 blocks of nc = 2) and at most NVT null vectors: ONE element
// per lane.  Phase 1 issues EVERY load of the site at once -- the KB fine values and all nvec null-vector entries of the
// lane's element, NVT + KB 16-byte loads in flight per lane, which is what hides the HBM latency at two wavefronts per
// SIMD -- and keeps them in registers.  Phase 2 runs nvec/DC passes out of registers: a pass forms DC = 16/KB null
// vectors x KB systems of products (32 doubles) and halves them over the 32 lanes in 16+8+4+2+1 = 31 exchanges down to
// ONE double per lane -- component (lane bit 4) of the sum for (d, system) = the bit-reversed low lane bits -- kept in a
// register per pass.  Phase 3 adds the nvec/DC results into the coarse vector (independent read-modify-writes).  While
// one wavefront of a SIMD computes, the other has its loads in flight.
template <typename T, int KB, int NVT>
__global__ __launch_bounds__(BLOCK) void k_brestrict_small(const void* __restrict__ nullv, int nvec, const void* __restrict__ fine, void* __restrict__ coarse,
                                                           const XferGeom g, const PassIds ids, long cstride, long fstride) {
  typedef typename CStore<T>::type ct;
  constexpr int DC = 16 / KB;        // null vectors per pass
  constexpr int NV = 32;             // doubles per lane per pass
  constexpr int NP = NVT / DC;       // passes
  const int ns = (ids.n < KB) ? ids.n : KB;
  const int cLx = 2 * g.chr;
  const long ncs = (long)cLx * g.cLy;
  const int G = (g.bx / 2) * g.fnc;
  const int nel = 2 * g.by * G;      // <= 32
  const int grp = threadIdx.x >> 5, l = threadIdx.x & 31;
  // the value this lane ends up holding after the halving: index bits (b0 b1 b2 b3 b4) = lane bits 0..4, MSB first
  const int idx = ((l & 1) << 4) | ((l & 2) << 2) | (l & 4) | ((l & 8) >> 2) | ((l & 16) >> 4);
  const int my_pair = idx >> 1, my_comp = idx & 1;
  const int my_dq = my_pair / KB, my_q = my_pair - my_dq * KB;
  for (long cs = (long)blockIdx.x * (BLOCK / 32) + grp; cs < ncs; cs += (long)gridDim.x * (BLOCK / 32)) {
    const int cy = (int)(cs / cLx), cx = (int)(cs - (long)cy * cLx);
    const long ci = coarse_site_index(g, cx, cy);
    const bool have = l < nel;
    const int rr = have ? l / G : 0, el = have ? l - rr * G : 0;
    const int p = rr / g.by, y = cy * g.by + (rr - p * g.by);
    const long e = ((long)p * g.fhalf_vol + (long)y * g.fhr + (long)cx * (g.bx / 2)) * g.fnc + el;
    // ---- phase 1: every load of this site.  UNCONDITIONAL loads from addresses that are always valid (a lane without an element reads the block's
    // first one, a null vector beyond nvec is read as vector 0, an unused system slot aliases slot 0) and are zeroed when they are used: a
    // divergent branch around each load made the compiler close the load with a full s_waitcnt at every join (21 of 48 loads waited for alone).
    const long e_safe = e;   // (rr = el = 0 for a lane without an element: the block's first element)
    ct fr[KB];
    typename RawC<T>::type nraw[NVT];   // (complex<float>: the raw 8 bytes -- splitting them behind each load made the loads wait for one another)
#pragma unroll
    for (int q = 0; q < KB; q++) fr[q] = reinterpret_cast<const ct*>(fine)[(long)ids.id[q] * fstride + e_safe];
#pragma unroll
    for (int d = 0; d < NVT; d++) nraw[d] = ld_raw_nt<T>(nullv, (long)(d < nvec ? d : 0) * g.fsize + e_safe);   // (complex<float>: ONE 8-byte load)
    {
      ct zero;
      zero.x = 0; zero.y = 0;
#pragma unroll
      for (int q = 0; q < KB; q++) if (!(have && q < ns)) fr[q] = zero;
    }
    // ---- phase 2: products and recursive halving, out of registers
    double res[NP];
#pragma unroll
    for (int ps = 0; ps < NP; ps++) {
      double v[NV];
#pragma unroll
      for (int dq = 0; dq < DC; dq++) {
        const cplx nw = widen_rawc<T>(nraw[ps * DC + dq]);
        const bool nval = have && (ps * DC + dq) < nvec;
        const double nx = nval ? nw.x : 0.0, ny = nval ? nw.y : 0.0;
#pragma unroll
        for (int q = 0; q < KB; q++) {   // conj(nv) f
          const double fx = (double)fr[q].x, fy = (double)fr[q].y;
          v[(dq * KB + q) * 2] = fma(nx, fx, ny * fy);
          v[(dq * KB + q) * 2 + 1] = fma(nx, fy, -ny * fx);
        }
      }
#pragma unroll
      for (int step = 0; step < 5; step++) {
        const int m = 1 << step;
        const int half = NV >> (step + 1);
        const bool up = (l & m) != 0;
#pragma unroll
        for (int i = 0; i < half; i++) {
          const double lo = v[i], hi = v[i + half];
          const double send = up ? lo : hi;
          const double keep = up ? hi : lo;
          double recv;
          if (m == 1) recv = lane_xor1(send);
          else if (m == 2) recv = lane_xor2(send);
          else recv = __shfl_xor(send, m);
          v[i] = keep + recv;
        }
      }
      res[ps] = v[0];
    }
    // ---- phase 3: one read-modify-write per pass (this lane's component of (d = ps*DC + my_dq, system my_q))
    if (my_q < ns) {
      T* out = reinterpret_cast<T*>(reinterpret_cast<ct*>(coarse) + (long)pick_id(ids, my_q) * cstride + ci * g.cnc) + my_comp;
      T prev[NP];
#pragma unroll
      for (int ps = 0; ps < NP; ps++) prev[ps] = (ps * DC + my_dq < nvec) ? out[2 * (ps * DC + my_dq)] : (T)0;
#pragma unroll
      for (int ps = 0; ps < NP; ps++)
        if (ps * DC + my_dq < nvec) out[2 * (ps * DC + my_dq)] = (T)((double)prev[ps] + res[ps]);
    }
  }
}

// two complex<float> per lane (16-byte accesses): even fnc (a pack stays inside one site), 16-byte aligned arrays
template <typename T>
static bool xfer_pack2(const XferGeom& g, const void* nullvecs, const void* fine) {
  return sizeof(T) == sizeof(float) && g_xfer_pack && !(g.fnc & 1) && !(g.fsize & 1) && aligned16(nullvecs) && aligned16(fine);
}

template <typename T>
static int launch_restrict(const void* nullvecs, int nvec, const void* fine, void* coarse, const XferGeom& g, hipStream_t st) {
  if ((g.bx & 1) == 0) {
    const bool pack2 = xfer_pack2<T>(g, nullvecs, fine);
    const int G = (g.bx / 2) * g.fnc / (pack2 ? 2 : 1);
    int NG = BLOCK / G;
    if (NG < 1) NG = 1;
    const int cLx = 2 * g.chr;
    if (NG > cLx) NG = cLx;
    const int TPG = BLOCK / NG;
    dim3 grid((unsigned)((cLx + NG - 1) / NG), g.cLy > 65535 ? 65535 : g.cLy);
    if (pack2) k_restrict<T, 2><<<grid, BLOCK, sizeof(cplx) * BLOCK * XFER_DC, st>>>(nullvecs, nvec, fine, coarse, g, NG, TPG);
    else k_restrict<T, 1><<<grid, BLOCK, sizeof(cplx) * BLOCK * XFER_DC, st>>>(nullvecs, nvec, fine, coarse, g, NG, TPG);
  } else {
    k_restrict_generic<T><<<grid_1d((size_t)4 * g.chalf_vol * nvec / 2), BLOCK, 0, st>>>(nullvecs, nvec, fine, coarse, g);
  }
  QMG_LAUNCH_CHECK();
  return QMG_SUCCESS;
}

template <typename T>
static int launch_prolong(const void* nullvecs, int nvec, const void* coarse, void* fine, const XferGeom& g, hipStream_t st) {
  const bool pack2 = xfer_pack2<T>(g, nullvecs, fine);
  const long row_packs = (long)g.fhr * g.fnc / (pack2 ? 2 : 1);
  unsigned gx = (unsigned)((row_packs + BLOCK - 1) / BLOCK);
  if (gx > 1024) gx = 1024;
  const int nrows = 2 * g.fLy;
  dim3 grid(gx, nrows > 65535 ? 65535 : nrows);
  if (pack2) k_prolong<T, 2><<<grid, BLOCK, 0, st>>>(nullvecs, nvec, coarse, fine, g);
  else k_prolong<T, 1><<<grid, BLOCK, 0, st>>>(nullvecs, nvec, coarse, fine, g);
  QMG_LAUNCH_CHECK();
  return QMG_SUCCESS;
}

// complex<double> vectors with complex<float> null vectors (a preconditioner level's narrow copy, TransferMG::enable_f32_shadow): two elements per lane,
// so that a lane's null-vector load stays 16 bytes.  Even fnc, even block width, 16-byte aligned null vectors.
static int launch_restrict_nv32(const void* null32, int nvec, const void* fine, void* coarse, const XferGeom& g, hipStream_t st) {
  if ((g.bx & 1) || (g.fnc & 1) || (g.fsize & 1) || !aligned16(null32)) return QMG_ERR_UNSUPPORTED;
  const int G = (g.bx / 2) * g.fnc / 2;
  int NG = BLOCK / (G > 0 ? G : 1);
  if (NG < 1) NG = 1;
  const int cLx = 2 * g.chr;
  if (NG > cLx) NG = cLx;
  const int TPG = BLOCK / NG;
  dim3 grid((unsigned)((cLx + NG - 1) / NG), g.cLy > 65535 ? 65535 : g.cLy);
  k_restrict<double, 2, float><<<grid, BLOCK, sizeof(cplx) * BLOCK * XFER_DC, st>>>(null32, nvec, fine, coarse, g, NG, TPG);
  QMG_LAUNCH_CHECK();
  return QMG_SUCCESS;
}
static int launch_prolong_nv32(const void* null32, int nvec, const void* coarse, void* fine, const XferGeom& g, hipStream_t st) {
  if ((g.fnc & 1) || (g.fsize & 1) || !aligned16(null32)) return QMG_ERR_UNSUPPORTED;
  const long row_packs = (long)g.fhr * g.fnc / 2;
  unsigned gx = (unsigned)((row_packs + BLOCK - 1) / BLOCK);
  if (gx > 1024) gx = 1024;
  const int nrows = 2 * g.fLy;
  dim3 grid(gx, nrows > 65535 ? 65535 : nrows);
  k_prolong<double, 2, float><<<grid, BLOCK, 0, st>>>(null32, nvec, coarse, fine, g);
  QMG_LAUNCH_CHECK();
  return QMG_SUCCESS;
}

// qmg_transfer_mfma.hip: the batched transfers as contractions on the matrix cores (SITE_DECLINED: shapes not served there)
int restrict_batch_mfma(int f32, const void* nullvecs, int nvec, const void* fine, void* coarse, int fhr, int fLy, int fnc, int chr, int cLy, int cnc, int bx, int by,
                        long fhalf_vol, long fsize, const int* ids8, int n, long cstride, long fstride, hipStream_t st);
int prolong_batch_mfma(int f32, const void* nullvecs, int nvec, const void* coarse, void* fine, int fhr, int fLy, int fnc, int chr, int cLy, int cnc, int bx, int by,
                       long fhalf_vol, long fsize, const int* ids8, int n, long cstride, long fstride, hipStream_t st);

int g_xfer_pack = 1;   // tuning knob "xfer_pack": complex<float> transfer kernels move two elements per lane (16-byte accesses)
int g_xfer_tile = 1;   // tuning knob "xfer_tile": 1 = batched transfer as LDS-tiled kernels, 0 = system by system

// sites per prolong tile: a fine half-row segment of at least 512 bytes where the lattice allows, LDS <= 48 KB
static int prolong_tile_sites(const XferGeom& g, int nvec, int KB) {
  const int G = (g.bx / 2) * g.fnc;
  int SX = (32 + G - 1) / G;
  if (SX < 1) SX = 1;
  if (SX > 2 * g.chr) SX = 2 * g.chr;
  while (SX > 1 && (size_t)SX * (nvec * KB + 1) * sizeof(cplx) > 48 * 1024) SX--;   // (sized for complex<double> entries; complex<float> tiles use half of it)
  return SX;
}

template <typename T>
static int prolong_batch_impl(const void* nullvecs, int nvec, const void* coarse, void* fine, const XferGeom& g, const BatchIdx& bi, size_t cstride,
                              size_t fstride, hipStream_t st) {
  typedef typename CStore<T>::type ct;
  if (bi.n == 1 || !g_xfer_tile || (g.bx & 1)) {   // one system (or an odd block width): the single-vector kernel, system by system
    for (int s = 0; s < bi.n; s++) {
      const int rc = launch_prolong<T>(nullvecs, nvec, (const ct*)coarse + (size_t)bi.id[s] * cstride, (ct*)fine + (size_t)bi.id[s] * fstride, g, st);
      if (rc) return rc;
    }
    return QMG_SUCCESS;
  }
  for (int s0 = 0; s0 < bi.n; s0 += 8) {
    const int left = bi.n - s0;
    {
      const PassIds pi = make_pass(bi, s0);
      const int rc = prolong_batch_mfma(sizeof(T) == 4, nullvecs, nvec, coarse, fine, g.fhr, g.fLy, g.fnc, g.chr, g.cLy, g.cnc, g.bx, g.by, g.fhalf_vol, g.fsize, pi.id, pi.n,
                                        (long)cstride, (long)fstride, st);
      if (rc != SITE_DECLINED) { if (rc) return rc; continue; }
    }
    const int KB = left > 4 ? 8 : left > 2 ? 4 : 2;
    const int SX = prolong_tile_sites(g, nvec, KB);
    const size_t smem = (size_t)SX * (nvec * KB + 1) * sizeof(ct);
    if (smem > 64 * 1024) return QMG_ERR_UNSUPPORTED;
    dim3 grid((unsigned)((2 * g.chr + SX - 1) / SX), g.cLy > 65535 ? 65535 : g.cLy);
    if (KB == 8 && sizeof(T) == 4 && nvec >= 12) k_bprolong_tile<T, 8, 12><<<grid, BLOCK, smem, st>>>(nullvecs, nvec, coarse, fine, g, make_pass(bi, s0), (long)cstride, (long)fstride, SX);
    else if (KB == 8) k_bprolong_tile<T, 8><<<grid, BLOCK, smem, st>>>(nullvecs, nvec, coarse, fine, g, make_pass(bi, s0), (long)cstride, (long)fstride, SX);
    else if (KB == 4) k_bprolong_tile<T, 4><<<grid, BLOCK, smem, st>>>(nullvecs, nvec, coarse, fine, g, make_pass(bi, s0), (long)cstride, (long)fstride, SX);
    else k_bprolong_tile<T, 2><<<grid, BLOCK, smem, st>>>(nullvecs, nvec, coarse, fine, g, make_pass(bi, s0), (long)cstride, (long)fstride, SX);
    QMG_LAUNCH_CHECK();
  }
  return QMG_SUCCESS;
}

template <typename T>
static int restrict_batch_impl(const void* nullvecs, int nvec, const void* fine, void* coarse, const XferGeom& g, const BatchIdx& bi, size_t fstride,
                               size_t cstride, hipStream_t st) {
  typedef typename CStore<T>::type ct;
  if (bi.n == 1 || !g_xfer_tile || (g.bx & 1)) {
    for (int s = 0; s < bi.n; s++) {
      const int rc = launch_restrict<T>(nullvecs, nvec, (const ct*)fine + (size_t)bi.id[s] * fstride, (ct*)coarse + (size_t)bi.id[s] * cstride, g, st);
      if (rc) return rc;
    }
    return QMG_SUCCESS;
  }
  const long ncs = 2 * g.chalf_vol;
  const long nblk = (ncs + BLOCK / 32 - 1) / (BLOCK / 32);
  const unsigned gx = (unsigned)(nblk > 262144 ? 262144 : nblk);
  const int nel = g.bx * g.by * g.fnc;
  for (int s0 = 0; s0 < bi.n; s0 += 8) {
    const int left = bi.n - s0;
    {
      const PassIds pi = make_pass(bi, s0);
      const int rc = restrict_batch_mfma(sizeof(T) == 4, nullvecs, nvec, fine, coarse, g.fhr, g.fLy, g.fnc, g.chr, g.cLy, g.cnc, g.bx, g.by, g.fhalf_vol, g.fsize, pi.id, pi.n,
                                         (long)cstride, (long)fstride, st);
      if (rc != SITE_DECLINED) { if (rc) return rc; continue; }
    }
    const int KB = left > 4 ? 8 : left > 2 ? 4 : 2;
    if (nel <= 32 && nvec <= 24 && g_xfer_tile != 2) {   // one element per lane, every load of a site in flight at once
#define QMG_RS(KBV, NVTV) k_brestrict_small<T, KBV, NVTV><<<gx, BLOCK, 0, st>>>(nullvecs, nvec, fine, coarse, g, make_pass(bi, s0), (long)cstride, (long)fstride)
#define QMG_RS_KB(KBV) { if (nvec <= 8) QMG_RS(KBV, 8); else if (nvec <= 16) QMG_RS(KBV, 16); else QMG_RS(KBV, 24); }
      if (KB == 8) QMG_RS_KB(8) else if (KB == 4) QMG_RS_KB(4) else QMG_RS_KB(2)
#undef QMG_RS_KB
#undef QMG_RS
      QMG_LAUNCH_CHECK();
      continue;
    }
    if (left > 4) k_brestrict_tile<T, 8><<<gx, BLOCK, 0, st>>>(nullvecs, nvec, fine, coarse, g, make_pass(bi, s0), (long)cstride, (long)fstride);
    else if (left > 2) k_brestrict_tile<T, 4><<<gx, BLOCK, 0, st>>>(nullvecs, nvec, fine, coarse, g, make_pass(bi, s0), (long)cstride, (long)fstride);
    else k_brestrict_tile<T, 2><<<gx, BLOCK, 0, st>>>(nullvecs, nvec, fine, coarse, g, make_pass(bi, s0), (long)cstride, (long)fstride);
    QMG_LAUNCH_CHECK();
  }
  return QMG_SUCCESS;
}

}  // namespace qmg

using namespace qmg;

extern "C" {

int qmg_prolong(const void* nullvecs, int nvec, const void* coarse, void* fine,
                int fLx, int fLy, int fnc, int cLx, int cLy, int cnc, void* stream) {
  if (!nullvecs || !coarse || !fine || nvec < 1 || nvec > cnc) return QMG_ERR_INVALID;
  XferGeom g;
  int rc = make_geom(&g, fLx, fLy, fnc, cLx, cLy, cnc);
  if (rc) return rc;
  return launch_prolong<double>(nullvecs, nvec, coarse, fine, g, as_stream(stream));
}

int qmg_restrict(const void* nullvecs, int nvec, const void* fine, void* coarse,
                 int fLx, int fLy, int fnc, int cLx, int cLy, int cnc, void* stream) {
  if (!nullvecs || !coarse || !fine || nvec < 1 || nvec > cnc) return QMG_ERR_INVALID;
  XferGeom g;
  int rc = make_geom(&g, fLx, fLy, fnc, cLx, cLy, cnc);
  if (rc) return rc;
  return launch_restrict<double>(nullvecs, nvec, fine, coarse, g, as_stream(stream));
}

// complex<double> vectors, complex<float> null vectors: system by system through the single-vector kernels (the facade sends ONE active system here;
// batches of several share one read of the fp64 null vectors in the tile kernels instead)
int qmg_prolong_batch_nv32(const void* null32, int nvec, const void* coarse, void* fine, int fLx, int fLy, int fnc, int cLx, int cLy, int cnc,
                           int nrhs, size_t cstride, size_t fstride, unsigned mask, void* stream) {
  if (!null32 || !coarse || !fine || nvec < 1 || nrhs < 1 || nrhs > BATCH_MAX) return QMG_ERR_INVALID;
  XferGeom g;
  int rc = make_geom(&g, fLx, fLy, fnc, cLx, cLy, cnc);
  if (rc) return rc;
  if (nvec != cnc) return QMG_ERR_INVALID;
  const BatchIdx bi = expand_mask(mask, nrhs);
  for (int s = 0; s < bi.n; s++) {
    rc = launch_prolong_nv32(null32, nvec, (const cplx*)coarse + (size_t)bi.id[s] * cstride, (cplx*)fine + (size_t)bi.id[s] * fstride, g, as_stream(stream));
    if (rc) return rc;
  }
  return QMG_SUCCESS;
}
int qmg_restrict_batch_nv32(const void* null32, int nvec, const void* fine, void* coarse, int fLx, int fLy, int fnc, int cLx, int cLy, int cnc,
                            int nrhs, size_t fstride, size_t cstride, unsigned mask, void* stream) {
  if (!null32 || !coarse || !fine || nvec < 1 || nrhs < 1 || nrhs > BATCH_MAX) return QMG_ERR_INVALID;
  XferGeom g;
  int rc = make_geom(&g, fLx, fLy, fnc, cLx, cLy, cnc);
  if (rc) return rc;
  if (nvec != cnc) return QMG_ERR_INVALID;
  const BatchIdx bi = expand_mask(mask, nrhs);
  for (int s = 0; s < bi.n; s++) {
    rc = launch_restrict_nv32(null32, nvec, (const cplx*)fine + (size_t)bi.id[s] * fstride, (cplx*)coarse + (size_t)bi.id[s] * cstride, g, as_stream(stream));
    if (rc) return rc;
  }
  return QMG_SUCCESS;
}

// transfer.h:455-511 for a lock-step batch, either storage precision
int qmg_prolong_batch_t(int dtype, const void* nullvecs, int nvec, const void* coarse, void* fine, int fLx, int fLy, int fnc, int cLx, int cLy, int cnc,
                        int nrhs, size_t cstride, size_t fstride, unsigned mask, void* stream) {
  if (!valid_dtype(dtype) || !nullvecs || !coarse || !fine || nvec < 1 || nrhs < 1 || nrhs > BATCH_MAX) return QMG_ERR_INVALID;
  XferGeom g;
  int rc = make_geom(&g, fLx, fLy, fnc, cLx, cLy, cnc);
  if (rc) return rc;
  if (nvec != cnc) return QMG_ERR_INVALID;
  const BatchIdx bi = expand_mask(mask, nrhs);
  if (bi.n == 0) return QMG_SUCCESS;
  if (dtype == QMG_C32) return prolong_batch_impl<float>(nullvecs, nvec, coarse, fine, g, bi, cstride, fstride, as_stream(stream));
  return prolong_batch_impl<double>(nullvecs, nvec, coarse, fine, g, bi, cstride, fstride, as_stream(stream));
}
int qmg_prolong_batch(const void* nullvecs, int nvec, const void* coarse, void* fine, int fLx, int fLy, int fnc, int cLx, int cLy, int cnc,
                      int nrhs, size_t cstride, size_t fstride, unsigned mask, void* stream) {
  return qmg_prolong_batch_t(QMG_C64, nullvecs, nvec, coarse, fine, fLx, fLy, fnc, cLx, cLy, cnc, nrhs, cstride, fstride, mask, stream);
}
int qmg_restrict_batch_t(int dtype, const void* nullvecs, int nvec, const void* fine, void* coarse, int fLx, int fLy, int fnc, int cLx, int cLy, int cnc,
                         int nrhs, size_t fstride, size_t cstride, unsigned mask, void* stream) {
  if (!valid_dtype(dtype) || !nullvecs || !coarse || !fine || nvec < 1 || nrhs < 1 || nrhs > BATCH_MAX) return QMG_ERR_INVALID;
  XferGeom g;
  int rc = make_geom(&g, fLx, fLy, fnc, cLx, cLy, cnc);
  if (rc) return rc;
  if (nvec != cnc) return QMG_ERR_INVALID;
  const BatchIdx bi = expand_mask(mask, nrhs);
  if (bi.n == 0) return QMG_SUCCESS;
  if (dtype == QMG_C32) return restrict_batch_impl<float>(nullvecs, nvec, fine, coarse, g, bi, fstride, cstride, as_stream(stream));
  return restrict_batch_impl<double>(nullvecs, nvec, fine, coarse, g, bi, fstride, cstride, as_stream(stream));
}
int qmg_restrict_batch(const void* nullvecs, int nvec, const void* fine, void* coarse, int fLx, int fLy, int fnc, int cLx, int cLy, int cnc,
                       int nrhs, size_t fstride, size_t cstride, unsigned mask, void* stream) {
  return qmg_restrict_batch_t(QMG_C64, nullvecs, nvec, fine, coarse, fLx, fLy, fnc, cLx, cLy, cnc, nrhs, fstride, cstride, mask, stream);
}

// block_orthonormalize (transfer.h:514-607): the reference's own formulation -- classical
// Gram-Schmidt per block phrased as single-vector restrict / prolong -- driven from the host,
// every pass a device kernel.  O(nvec^2) launches; setup-time only.
}  // extern "C"
namespace qmg {
// (fallback of qmg_block_orthonormalize_n, csrc/qmg_setup.hip: odd block widths, tiles beyond LDS, "setup_fused" 0)
int block_orthonormalize_passes(void* nullvecs, int nvec, int fLx, int fLy, int fnc, int cLx, int cLy,
                                void* cholesky, void* stream) {
  if (!nullvecs || nvec < 1) return QMG_ERR_INVALID;
  XferGeom g;
  const int cnc = nvec;
  int rc = make_geom(&g, fLx, fLy, fnc, cLx, cLy, cnc);
  if (rc) return rc;
  hipStream_t st = as_stream(stream);
  const long fsize = g.fsize, cvol = 2 * g.chalf_vol, csize = cvol * cnc;
  cplx* nv = (cplx*)nullvecs;
  cplx *fine1 = nullptr, *coarse2 = nullptr;
  QMG_HIP_CHECK(hipMalloc((void**)&fine1, sizeof(cplx) * fsize));
  QMG_HIP_CHECK(hipMalloc((void**)&coarse2, sizeof(cplx) * csize));
  rc = QMG_SUCCESS;
  for (int i = 0; i < nvec && !rc; i++) {
    for (int j = 0; j < i && !rc; j++) {
      hipMemsetAsync(fine1, 0, sizeof(cplx) * fsize, st);
      hipMemsetAsync(coarse2, 0, sizeof(cplx) * csize, st);
      rc = launch_restrict<double>(nv + j * fsize, 1, nv + i * fsize, coarse2, g, st);            // <v_i, v_j> per block (:552)
      if (!rc && cholesky) k_chol_store<<<grid_1d((size_t)cvol), BLOCK, 0, st>>>((cplx*)cholesky, coarse2, cvol, cnc, j * cnc + i, 0);   // :560
      if (!rc) rc = launch_prolong<double>(nv + j * fsize, 1, coarse2, fine1, g, st);             // <v_i, v_j> v_j (:565)
      if (!rc) rc = qmg_caxpy(-1.0, 0.0, fine1, nv + i * fsize, (size_t)fsize, stream);   // :569
    }
    if (rc) break;
    hipMemsetAsync(fine1, 0, sizeof(cplx) * fsize, st);
    hipMemsetAsync(coarse2, 0, sizeof(cplx) * csize, st);
    rc = launch_restrict<double>(nv + i * fsize, 1, nv + i * fsize, coarse2, g, st);              // :579
    if (rc) break;
    k_inv_real_sqrt<<<grid_1d((size_t)csize), BLOCK, 0, st>>>(coarse2, csize);            // :583
    if (cholesky) k_chol_store<<<grid_1d((size_t)cvol), BLOCK, 0, st>>>((cplx*)cholesky, coarse2, cvol, cnc, i * (cnc + 1), 1);   // :588-593
    rc = launch_prolong<double>(nv + i * fsize, 1, coarse2, fine1, g, st);                        // :598
    if (rc) break;
    hipMemcpyAsync(nv + i * fsize, fine1, sizeof(cplx) * fsize, hipMemcpyDeviceToDevice, st);   // :601
  }
  hipStreamSynchronize(st);
  hipFree(fine1);
  hipFree(coarse2);
  if (!rc) QMG_LAUNCH_CHECK();
  return rc;
}

}  // namespace qmg
extern "C" {

// block_bi_orthonormalize, one pass, in place (transfer.h:610-769): the asymmetric (P != R^dag) counterpart.
// pvecs = prolongator vectors, rvecs = restrictor vectors; afterwards R^dag P = 1 block by block.
// block_L / block_U (coarse size_cm each) may be NULL.
int qmg_block_bi_orthonormalize(void* pvecs, void* rvecs, int nvec, int fLx, int fLy, int fnc, int cLx, int cLy,
                                void* block_L, void* block_U, void* stream) {
  if (!pvecs || !rvecs || nvec < 1) return QMG_ERR_INVALID;
  XferGeom g;
  const int cnc = nvec;
  int rc = make_geom(&g, fLx, fLy, fnc, cLx, cLy, cnc);
  if (rc) return rc;
  hipStream_t st = as_stream(stream);
  const long fsize = g.fsize, cvol = 2 * g.chalf_vol, csize = cvol * cnc;
  cplx* P = (cplx*)pvecs;
  cplx* R = (cplx*)rvecs;
  cplx *fine1 = nullptr, *coarse2 = nullptr;
  QMG_HIP_CHECK(hipMalloc((void**)&fine1, sizeof(cplx) * fsize));
  QMG_HIP_CHECK(hipMalloc((void**)&coarse2, sizeof(cplx) * csize));
  auto zero = [&]() { hipMemsetAsync(fine1, 0, sizeof(cplx) * fsize, st); hipMemsetAsync(coarse2, 0, sizeof(cplx) * csize, st); };
  rc = QMG_SUCCESS;
  for (int i = 0; i < nvec && !rc; i++) {
    for (int j = 0; j < i && !rc; j++) {
      zero();
      rc = launch_restrict<double>(R + j * fsize, 1, P + i * fsize, coarse2, g, st);                                        // <r_j, p_i>  (:643)
      if (!rc && block_U) k_chol_store<<<grid_1d((size_t)cvol), BLOCK, 0, st>>>((cplx*)block_U, coarse2, cvol, cnc, j * cnc + i, 0);   // :651
      if (!rc) rc = launch_prolong<double>(P + j * fsize, 1, coarse2, fine1, g, st);                                        // :656
      if (!rc) rc = qmg_caxpy(-1.0, 0.0, fine1, P + i * fsize, (size_t)fsize, stream);                              // :660
      if (rc) break;
      zero();
      rc = launch_restrict<double>(P + j * fsize, 1, R + i * fsize, coarse2, g, st);                                        // <p_j, r_i>  (:668)
      if (!rc && block_L) k_chol_store<<<grid_1d((size_t)cvol), BLOCK, 0, st>>>((cplx*)block_L, coarse2, cvol, cnc, i * cnc + j, 0);   // :678
      if (!rc) rc = launch_prolong<double>(R + j * fsize, 1, coarse2, fine1, g, st);                                        // :683
      if (!rc) rc = qmg_caxpy(-1.0, 0.0, fine1, R + i * fsize, (size_t)fsize, stream);                              // :687
    }
    if (rc) break;
    zero();
    rc = launch_restrict<double>(R + i * fsize, 1, P + i * fsize, coarse2, g, st);                                          // <r_i, p_i>  (:699)
    if (rc) break;
    k_elementwise<0><<<grid_1d((size_t)csize), BLOCK, 0, st>>>(coarse2, csize);                                     // inv_phase_abs_sqrt (:703)
    if (block_L) k_chol_store<<<grid_1d((size_t)cvol), BLOCK, 0, st>>>((cplx*)block_L, coarse2, cvol, cnc, i * (cnc + 1), 1);   // :708-719
    rc = launch_prolong<double>(R + i * fsize, 1, coarse2, fine1, g, st);                                                   // :724
    if (rc) break;
    hipMemcpyAsync(R + i * fsize, fine1, sizeof(cplx) * fsize, hipMemcpyDeviceToDevice, st);                        // :727
    hipMemsetAsync(fine1, 0, sizeof(cplx) * fsize, st);
    k_elementwise<1><<<grid_1d((size_t)csize), BLOCK, 0, st>>>(coarse2, csize);                                     // abs_vector (:731)
    if (block_U) k_chol_store<<<grid_1d((size_t)cvol), BLOCK, 0, st>>>((cplx*)block_U, coarse2, cvol, cnc, i * (cnc + 1), 1);   // :735-742
    rc = launch_prolong<double>(P + i * fsize, 1, coarse2, fine1, g, st);                                                   // :747
    if (rc) break;
    hipMemcpyAsync(P + i * fsize, fine1, sizeof(cplx) * fsize, hipMemcpyDeviceToDevice, st);                        // :750
  }
  if (!rc && block_L) k_elementwise<2><<<grid_1d((size_t)cvol * cnc * cnc), BLOCK, 0, st>>>((cplx*)block_L, cvol * cnc * cnc);   // conj L (:757)
  hipStreamSynchronize(st);
  hipFree(fine1);
  hipFree(coarse2);
  if (!rc) QMG_LAUNCH_CHECK();
  return rc;
}

// CoarseOperator2D ctor, steps 1-2 (coarse.h:137-444): 9 probes per coarse colour, each
// unit vector -> prolong -> partial fine apply -> restrict -> scatter into column `color`.
}  // extern "C"
namespace qmg {
// (fallback of qmg_coarse_build, csrc/qmg_setup.hip: more than 32 coarse colours, fine blocks beyond LDS, "setup_fused" 0)
int coarse_build_probes(void* cclover, void* chopping, const qmg_stencil_desc* fine, const void* nullvecs,
                        const void* restrict_vecs, int cLx, int cLy, int cnc, void* stream) {
  if (!cclover || !chopping || !fine || !nullvecs) return QMG_ERR_INVALID;
  XferGeom g;
  int rc = make_geom(&g, fine->Lx, fine->Ly, fine->nc, cLx, cLy, cnc);
  if (rc) return rc;
  hipStream_t st = as_stream(stream);
  const long fsize = g.fsize, cvol = 2 * g.chalf_vol, csize = cvol * cnc, ccm = csize * cnc;
  const void* rvecs = restrict_vecs ? restrict_vecs : nullvecs;
  cplx *tc = nullptr, *tf = nullptr, *taf = nullptr;
  QMG_HIP_CHECK(hipMalloc((void**)&tc, sizeof(cplx) * csize));
  QMG_HIP_CHECK(hipMalloc((void**)&tf, sizeof(cplx) * fsize));
  QMG_HIP_CHECK(hipMalloc((void**)&taf, sizeof(cplx) * fsize));
  hipMemsetAsync(cclover, 0, sizeof(cplx) * ccm, st);
  hipMemsetAsync(chopping, 0, sizeof(cplx) * 4 * ccm, st);
  auto probe = [&](int color, long lo, long hi, unsigned pieces) -> int {
    k_unit_probe<<<grid_1d((size_t)csize), BLOCK, 0, st>>>(tc, cvol, cnc, color, lo, hi);
    hipMemsetAsync(tf, 0, sizeof(cplx) * fsize, st);
    int r = launch_prolong<double>(nullvecs, cnc, tc, tf, g, st);
    if (r) return r;
    r = qmg_stencil_apply(fine, taf, tf, pieces | QMG_P_ZERO, 1, 0, stream);
    if (r) return r;
    hipMemsetAsync(tc, 0, sizeof(cplx) * csize, st);
    return launch_restrict<double>(rvecs, cnc, taf, tc, g, st);
  };
  rc = QMG_SUCCESS;
  for (int color = 0; color < cnc && !rc; color++) {
    rc = probe(color, 0, cvol, QMG_P_CLOVER);
    if (rc) break;
    k_probe_scatter<<<grid_1d((size_t)csize), BLOCK, 0, st>>>((cplx*)cclover, nullptr, tc, cvol, cnc, color, 0, cvol, 1);
    for (int dir = 0; dir < 4 && !rc; dir++) {
      const unsigned pieces = (QMG_P_EO_XP1 << dir) | (QMG_P_OE_XP1 << dir);
      const int fold = ((dir & 1) == 0) ? (cLx == 1) : (cLy == 1);
      for (int par = 0; par < 2 && !rc; par++) {
        const long lo = par * (cvol / 2), hi = lo + cvol / 2;
        rc = probe(color, lo, hi, pieces);
        if (rc) break;
        k_probe_scatter<<<grid_1d((size_t)csize), BLOCK, 0, st>>>((cplx*)cclover, (cplx*)chopping + dir * ccm, tc, cvol, cnc, color, lo, hi, fold);
      }
    }
  }
  hipStreamSynchronize(st);
  hipFree(tc);
  hipFree(tf);
  hipFree(taf);
  if (!rc) QMG_LAUNCH_CHECK();
  return rc;
}

}  // namespace qmg

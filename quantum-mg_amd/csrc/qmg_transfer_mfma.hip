// qmg_transfer_mfma.hip -- restrict / prolong of a lock-step batch as a CONTRACTION on the matrix cores (transfer/transfer.h:455-511 for
// up to 8 systems per pass).
//
// For one coarse site the batched transfer is a small dense product over the block's nel fine elements:
//   restrict   C[d][q]  = sum_e conj(N[d][e]) F[e][q]        (nvec x nel) . (nel x k)
//   prolong    F[e][q] += sum_d N[d][e] C[d][q]              (nel x nvec) . (nvec x k)
// with N the null vectors (read once, the bulk of the bytes: nvec of the nvec + 2k vector passes), F the fine and C the coarse values of
// the k systems.  Arithmetic intensity k/(1 + 2k/nvec) complex MACs per loaded element: at k = 8 the vector-FMA kernels of
// qmg_transfer.hip need 64 FMA-pairs per 8 (fp32) or 16 (fp64) bytes and were bound by their cross-lane sums (restrict: 0.20-0.26 of the
// HBM rate in fp32, 0.56 in fp64) or their LDS broadcasts (prolong: 0.42-0.48 in fp32) -- a real contraction, so it goes to MFMA
// (v_mfma_f64_16x16x4_f64 for BOTH storage precisions: complex<float> operands are widened when they are pulled from LDS, so the sums are
// formed in fp64 and rounded once, as in every other fp32-storage kernel of the library -- at 8 systems the f64 matrix pipe needs ~60 % of
// the time the null-vector stream takes), which leaves the kernels with that stream as their only cost.
//
// Real form with the k <= 8 systems' real and imaginary parts as the 16 MFMA columns (restrict) or rows (prolong):
//   restrict   columns [Fr_0..7 | Fi_0..7];  P = Nr . [Fr|Fi],  Q = Ni . [Fr|Fi];   Cr = P[:, q] + Q[:, q+8],  Ci = P[:, q+8] - Q[:, q]
//   prolong    rows    [Cr_0..7 ; Ci_0..7];  P = [Cr;Ci] . Nr,  Q = [Cr;Ci] . Ni;   Fr = P[q] - Q[q+8],        Fi = Q[q] + P[q+8]
// so a complex product costs TWO real MFMAs per tile.  One wavefront owns one coarse site; a workgroup owns SX = 4 consecutive coarse sites
// of a coarse row, whose fine elements on one fine half-row are CONTIGUOUS (SX G elements, G = (bx/2) nc_f): the workgroup stages a chunk
// of the null vectors [nvec][rows of the chunk][SX G] and of the fine vectors in LDS with whole-line coalesced loads and the wavefronts
// pull their MFMA operands from there (row strides odd in 8-byte words: the 16 lanes of an operand column hit 16 different LDS words).
// The prolong writes its result tile back through LDS, so the read-modify-write of the fine vectors is whole lines as well.
// Operand maps (as kernel C of qmg_stencil.hip): lane = 16 lq + lr; A (16 x 4): row lr, k lq; B (4 x 16): k lq, column lr;
// C (16 x 16): column lr, row 4 i + lq in accumulator register i.
#include "qmg_common.h"

namespace qmg {

// tuning knob "xfer_mfma": 1 (default) = the matrix cores where they are the faster kernel -- the complex<float> restrict from the fine level with 5-8
// systems (restrict_batch_mfma below: the list shrank when the vector-FMA restricts were fixed); the fp64 restrict and both prolongs stay with the
// vector-FMA kernels, which are at or above the MFMA form there;
// 2 = every shape the MFMA kernels serve (measurements); 0 = never
int g_xfer_mfma = 1;

struct XferGeomM {      // (the geometry of qmg_transfer.hip, restated: the two files share no header beyond qmg_common.h)
  int fhr, fLy, fnc, chr, cLy, cnc, bx, by;
  long fhalf_vol, fsize;
};
struct PassIdsM { int n; int id[8]; };
struct MfmaTile { int SX, CR, nchunk, Dstride, Fstride, G, R; };

__device__ __forceinline__ long m_coarse_site_index(const XferGeomM& g, int cx, int cy) {
  const int p = (cx + cy) & 1;
  return (long)(cy + p * g.cLy) * g.chr + (cx >> 1);
}
__device__ __forceinline__ int m_pick_id(const PassIdsM& p, int q) {   // run-time slot -> system id without indexing the by-value struct
  const int a = (q & 1) ? p.id[1] : p.id[0], b = (q & 1) ? p.id[3] : p.id[2], c = (q & 1) ? p.id[5] : p.id[4], d = (q & 1) ? p.id[7] : p.id[6];
  const int ab = (q & 2) ? b : a, cd = (q & 2) ? d : c;
  return (q & 4) ? cd : ab;
}

typedef double v4d __attribute__((ext_vector_type(4)));
__device__ __forceinline__ v4d mfma64(double a, double b, v4d c) { return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0); }

// first fine element of half-row `rr` (0 .. 2 by - 1: parity-major) of the block row cy, at the column where coarse site cx0 starts
__device__ __forceinline__ long m_run_base(const XferGeomM& g, int cy, int rr, int cx0) {
  const int p = rr / g.by, y = cy * g.by + (rr - p * g.by);
  return ((long)p * g.fhalf_vol + (long)y * g.fhr + (long)cx0 * (g.bx / 2)) * g.fnc;
}

// ---------------------------------------------------------------------------------------------------------------------
// restrict: coarse[q][ci][d] += sum_e conj(null[d][e]) fine[q][e]
// ---------------------------------------------------------------------------------------------------------------------
template <typename T, int MT>
__global__ __launch_bounds__(BLOCK) void k_brestrict_mfma(const void* __restrict__ nullv, int nvec, const void* __restrict__ fine, void* __restrict__ coarse,
                                                          const XferGeomM g, const PassIdsM ids, long cstride, long fstride, const MfmaTile L) {
  typedef typename CStore<T>::type ct;
  extern __shared__ __align__(16) unsigned char smem_raw[];
  ct* nt = reinterpret_cast<ct*>(smem_raw);                  // [nvec][Dstride]
  ct* ft = nt + (size_t)nvec * L.Dstride;                    // [8][Fstride]
  const int tid = threadIdx.x, lane = tid & (WAVE - 1), wv = tid / WAVE;
  const int lr = lane & 15, lq = lane >> 4;
  const int ns = ids.n;
  const int cLx = 2 * g.chr;
  const int cx0 = blockIdx.x * L.SX;
  const int nsx = (cLx - cx0 < L.SX) ? cLx - cx0 : L.SX;
  const int G = L.G, rowlen = nsx * G;                       // elements of the workgroup's sites on one fine half-row
  const ct* nul = reinterpret_cast<const ct*>(nullv);
  const ct* fin = reinterpret_cast<const ct*>(fine);
  ct zero; zero.x = 0; zero.y = 0;
  for (int cy = blockIdx.y; cy < g.cLy; cy += gridDim.y) {
    v4d P[MT], Q[MT];
#pragma unroll
    for (int t = 0; t < MT; t++) { P[t] = (v4d)(0); Q[t] = (v4d)(0); }
    for (int c = 0; c < L.nchunk; c++) {
      const int rr0 = c * L.CR;
      __syncthreads();   // the previous chunk's operand reads are done
      // ---- stage the chunk: null vectors [d][rl][u], fine vectors [q][rl][u]; u runs over the contiguous rowlen elements of a half-row.
      // A thread owns (rl, u) positions and walks the vectors at them: the index arithmetic (two divisions) is done once per position, the
      // loop over d / q only adds strides (the first version divided per element and was bound by its address arithmetic: 21 us per tile).
      {
        const int pairs = L.CR * rowlen;
        const int dgroups = (pairs < BLOCK) ? BLOCK / pairs : 1;
        const int my_dg = (pairs < BLOCK) ? tid / pairs : 0;
        const int pstep = (pairs < BLOCK) ? pairs : BLOCK;
        if (my_dg < dgroups)
          for (int pr = (pairs < BLOCK) ? tid - my_dg * pairs : tid; pr < pairs; pr += pstep) {
            const int rl = pr / rowlen, u = pr - rl * rowlen;
            const long gbase = m_run_base(g, cy, rr0 + rl, cx0) + u;
            const int lidx = rl * (L.SX * G) + u;
            // batches of 8 loads in flight per thread, then the 8 LDS stores (a load -> store loop is one memory latency per element)
            for (int d0 = my_dg; d0 < nvec; d0 += 8 * dgroups) {
              ct v[8];
#pragma unroll
              for (int k = 0; k < 8; k++) { const int d = d0 + k * dgroups; v[k] = zero; if (d < nvec) v[k] = nul[(long)d * g.fsize + gbase]; }
#pragma unroll
              for (int k = 0; k < 8; k++) { const int d = d0 + k * dgroups; if (d < nvec) nt[(size_t)d * L.Dstride + lidx] = v[k]; }
            }
            {
              ct v[8];
#pragma unroll
              for (int k = 0; k < 8; k++) { const int q = my_dg + k * dgroups; v[k] = zero; if (q < ns) v[k] = fin[(long)m_pick_id(ids, q) * fstride + gbase]; }
#pragma unroll
              for (int k = 0; k < 8; k++) { const int q = my_dg + k * dgroups; if (q < 8) ft[(size_t)q * L.Fstride + lidx] = v[k]; }
            }
          }
      }
      __syncthreads();
      if (wv < nsx) {
        const int ksteps = L.CR * G / 4;
        const int q = lr & 7, comp = lr >> 3;
        for (int ks = 0; ks < ksteps; ks++) {
          const int kk = 4 * ks + lq;
          const int rl = kk / G, el = kk - rl * G;
          const int u = rl * (L.SX * G) + wv * G + el;
          const ct fv = ft[(size_t)q * L.Fstride + u];
          const double b = comp ? (double)fv.y : (double)fv.x;
#pragma unroll
          for (int t = 0; t < MT; t++) {
            const int d = 16 * t + lr;
            ct nv = nt[(size_t)(d < nvec ? d : 0) * L.Dstride + u];   // (a select between a load and a constant struct becomes a select of addresses)
            if (d >= nvec) { nv.x = 0; nv.y = 0; }
            P[t] = mfma64((double)nv.x, b, P[t]);
            Q[t] = mfma64((double)nv.y, b, Q[t]);
          }
        }
      }
    }
    // ---- epilogue: column lr = q (re part of system q) for lr < 8, q + 8 (im part) otherwise; partner column lr ^ 8
    if (wv < nsx) {
      const long ci = m_coarse_site_index(g, cx0 + wv, cy);
      ct* cor = reinterpret_cast<ct*>(coarse);
#pragma unroll
      for (int t = 0; t < MT; t++)
#pragma unroll
        for (int i = 0; i < 4; i++) {
          const double px = __shfl_xor(P[t][i], 8), qx = __shfl_xor(Q[t][i], 8);
          const int d = 16 * t + 4 * i + lq;
          if (lr < 8 && lr < ns && d < nvec) {
            const double re = P[t][i] + qx, im = px - Q[t][i];     // Cr = P[q] + Q[q+8], Ci = P[q+8] - Q[q]
            // (read here, not before the tile loop: held across it the eight values cost restrict<float, 2> a wavefront of occupancy -- 155 VGPRs --
            // and 2048^2 -> 512^2 x 24 went from 1.10 back to 1.49 ms)
            ct* o = cor + (long)m_pick_id(ids, lr) * cstride + ci * g.cnc + d;
            const ct v = *o;
            ct w;
            w.x = (T)((double)v.x + re); w.y = (T)((double)v.y + im);
            *o = w;
          }
        }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// prolong: fine[q][e] += sum_d null[d][e] coarse[q][ci][d]
// ---------------------------------------------------------------------------------------------------------------------
template <typename T, int KS>   // KS = k-steps = ceil(nvec / 4)
__global__ __launch_bounds__(BLOCK) void k_bprolong_mfma(const void* __restrict__ nullv, int nvec, const void* __restrict__ coarse, void* __restrict__ fine,
                                                         const XferGeomM g, const PassIdsM ids, long cstride, long fstride, const MfmaTile L) {
  typedef typename CStore<T>::type ct;
  extern __shared__ __align__(16) unsigned char smem_raw[];
  ct* nt = reinterpret_cast<ct*>(smem_raw);                  // [nvec][Dstride]
  ct* ot = nt + (size_t)nvec * L.Dstride;                    // [8][Fstride]: the result tile
  const int tid = threadIdx.x, lane = tid & (WAVE - 1), wv = tid / WAVE;
  const int lr = lane & 15, lq = lane >> 4;
  const int ns = ids.n;
  const int cLx = 2 * g.chr;
  const int cx0 = blockIdx.x * L.SX;
  const int nsx = (cLx - cx0 < L.SX) ? cLx - cx0 : L.SX;
  const int G = L.G, rowlen = nsx * G;
  const ct* nul = reinterpret_cast<const ct*>(nullv);
  const ct* cor = reinterpret_cast<const ct*>(coarse);
  ct* fin = reinterpret_cast<ct*>(fine);
  ct zero; zero.x = 0; zero.y = 0;
  for (int cy = blockIdx.y; cy < g.cLy; cy += gridDim.y) {
    // A operand for every k-step: row lr = (system lr & 7, component lr >> 3), k = d = 4 ks + lq -- this site's coarse values
    double av[KS];
    {
      const int q = lr & 7, comp = lr >> 3;
      const long ci = (wv < nsx) ? m_coarse_site_index(g, cx0 + wv, cy) : 0;
#pragma unroll
      for (int ks = 0; ks < KS; ks++) {
        const int d = 4 * ks + lq;
        ct cv = zero;
        if (wv < nsx && q < ns && d < nvec) cv = cor[(long)m_pick_id(ids, q) * cstride + ci * g.cnc + d];
        av[ks] = comp ? (double)cv.y : (double)cv.x;
      }
    }
    for (int c = 0; c < L.nchunk; c++) {
      const int rr0 = c * L.CR;
      __syncthreads();   // the previous chunk's tile has been written back
      const int pairs = L.CR * rowlen;
      const int dgroups = (pairs < BLOCK) ? BLOCK / pairs : 1;
      const int my_dg = (pairs < BLOCK) ? tid / pairs : 0;
      const int pstep = (pairs < BLOCK) ? pairs : BLOCK;
      const int pr0 = (pairs < BLOCK) ? tid - my_dg * pairs : tid;
      if (my_dg < dgroups)
        for (int pr = pr0; pr < pairs; pr += pstep) {
          const int rl = pr / rowlen, u = pr - rl * rowlen;
          const long gbase = m_run_base(g, cy, rr0 + rl, cx0) + u;
          const int lidx = rl * (L.SX * G) + u;
          for (int d0 = my_dg; d0 < nvec; d0 += 8 * dgroups) {   // 8 loads in flight per thread, then the 8 LDS stores
            ct v[8];
#pragma unroll
            for (int k = 0; k < 8; k++) { const int d = d0 + k * dgroups; v[k] = zero; if (d < nvec) v[k] = nul[(long)d * g.fsize + gbase]; }
#pragma unroll
            for (int k = 0; k < 8; k++) { const int d = d0 + k * dgroups; if (d < nvec) nt[(size_t)d * L.Dstride + lidx] = v[k]; }
          }
        }
      __syncthreads();
      if (wv < nsx) {
        const int etiles = L.CR * G / 16;
        for (int et = 0; et < etiles; et++) {
          const int e = 16 * et + lr;                       // B column: this lane's element of the site's chunk
          const int rl = e / G, el = e - rl * G;
          const int u = rl * (L.SX * G) + wv * G + el;
          v4d P = (v4d)(0), Q = (v4d)(0);
#pragma unroll
          for (int ks = 0; ks < KS; ks++) {
            const int d = 4 * ks + lq;
            ct nv = nt[(size_t)(d < nvec ? d : 0) * L.Dstride + u];   // (a select between a load and a constant struct becomes a select of addresses)
            if (d >= nvec) { nv.x = 0; nv.y = 0; }
            P = mfma64(av[ks], (double)nv.x, P);
            Q = mfma64(av[ks], (double)nv.y, Q);
          }
          // rows q (re row of system q) and q + 8 (im row) are accumulator registers i and i + 2 of the SAME lane (row 4 i + lq)
#pragma unroll
          for (int i = 0; i < 2; i++) {
            const int q = 4 * i + lq;
            ct v;
            v.x = (T)(P[i] - Q[i + 2]); v.y = (T)(Q[i] + P[i + 2]);     // Fr = P[q] - Q[q+8], Fi = Q[q] + P[q+8]
            ot[(size_t)q * L.Fstride + u] = v;
          }
        }
      }
      __syncthreads();
      // ---- fine += tile, whole lines (same ownership of positions as the staging)
      if (my_dg < dgroups)
        for (int pr = pr0; pr < pairs; pr += pstep) {
          const int rl = pr / rowlen, u = pr - rl * rowlen;
          const long gbase = m_run_base(g, cy, rr0 + rl, cx0) + u;
          const int lidx = rl * (L.SX * G) + u;
          ct v[8];
#pragma unroll
          for (int k = 0; k < 8; k++) { const int q = my_dg + k * dgroups; v[k] = zero; if (q < ns) v[k] = fin[(long)m_pick_id(ids, q) * fstride + gbase]; }
#pragma unroll
          for (int k = 0; k < 8; k++) {
            const int q = my_dg + k * dgroups;
            if (q < ns) {
              const ct a = ot[(size_t)q * L.Fstride + lidx];
              ct w;
              w.x = (T)((double)v[k].x + (double)a.x); w.y = (T)((double)v[k].y + (double)a.y);
              fin[(long)m_pick_id(ids, q) * fstride + gbase] = w;
            }
          }
        }
    }
  }
}

// tile shape for a geometry, or SX = 0 when the matrix-core kernels do not serve it
static MfmaTile make_mfma_tile(const XferGeomM& g, int nvec, size_t esz, bool prolong) {
  MfmaTile L;
  L.SX = 0;
  if ((g.bx & 1) || nvec > 32) return L;
  L.G = (g.bx / 2) * g.fnc;
  L.R = 2 * g.by;
  const int need = prolong ? 16 : 4;                     // elements of a site per chunk: a multiple of the MFMA's N (prolong) / K (restrict) extent
  int best = 0;
  for (int cr = 1; cr <= L.R; cr++) {
    if (L.R % cr || (cr * L.G) % need) continue;
    const size_t bytes = (size_t)(nvec + 8) * ((size_t)cr * 4 * L.G + 1) * esz;
    if (bytes <= 60 * 1024) best = cr;
  }
  if (!best) return L;
  L.SX = 4;
  L.CR = best;
  L.nchunk = L.R / best;
  L.Dstride = L.CR * L.SX * L.G + 1;                     // odd in elements: operand columns spread over the LDS words
  L.Fstride = L.CR * L.SX * L.G + 1;
  return L;
}

// C-linkage-free entry points for qmg_transfer.hip: SITE_DECLINED when the shapes are not served
template <typename T>
static int restrict_mfma_t(const void* nullvecs, int nvec, const void* fine, void* coarse, const XferGeomM& g, const PassIdsM& ids, long cstride, long fstride, hipStream_t st) {
  const MfmaTile L = make_mfma_tile(g, nvec, 2 * sizeof(T), false);
  if (!L.SX) return SITE_DECLINED;
  const size_t smem = ((size_t)nvec * L.Dstride + (size_t)8 * L.Fstride) * 2 * sizeof(T);
  dim3 grid((unsigned)((2 * g.chr + L.SX - 1) / L.SX), g.cLy > 65535 ? 65535u : (unsigned)g.cLy);
  if (nvec <= 16) {
    if (smem > 64 * 1024) QMG_HIP_CHECK(hipFuncSetAttribute((const void*)k_brestrict_mfma<T, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
    k_brestrict_mfma<T, 1><<<grid, BLOCK, smem, st>>>(nullvecs, nvec, fine, coarse, g, ids, cstride, fstride, L);
  } else {
    if (smem > 64 * 1024) QMG_HIP_CHECK(hipFuncSetAttribute((const void*)k_brestrict_mfma<T, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
    k_brestrict_mfma<T, 2><<<grid, BLOCK, smem, st>>>(nullvecs, nvec, fine, coarse, g, ids, cstride, fstride, L);
  }
  QMG_LAUNCH_CHECK();
  return QMG_SUCCESS;
}
template <typename T>
static int prolong_mfma_t(const void* nullvecs, int nvec, const void* coarse, void* fine, const XferGeomM& g, const PassIdsM& ids, long cstride, long fstride, hipStream_t st) {
  const MfmaTile L = make_mfma_tile(g, nvec, 2 * sizeof(T), true);
  if (!L.SX) return SITE_DECLINED;
  const size_t smem = ((size_t)nvec * L.Dstride + (size_t)8 * L.Fstride) * 2 * sizeof(T);
  dim3 grid((unsigned)((2 * g.chr + L.SX - 1) / L.SX), g.cLy > 65535 ? 65535u : (unsigned)g.cLy);
  const int ks = (nvec + 3) / 4;
#define QMG_PM(KSV)                                                                                                                                   \
  {                                                                                                                                                   \
    if (smem > 64 * 1024) QMG_HIP_CHECK(hipFuncSetAttribute((const void*)k_bprolong_mfma<T, KSV>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem)); \
    k_bprolong_mfma<T, KSV><<<grid, BLOCK, smem, st>>>(nullvecs, nvec, coarse, fine, g, ids, cstride, fstride, L);                                      \
  }
  if (ks <= 2) QMG_PM(2) else if (ks <= 4) QMG_PM(4) else if (ks <= 6) QMG_PM(6) else QMG_PM(8)
#undef QMG_PM
  QMG_LAUNCH_CHECK();
  return QMG_SUCCESS;
}

int restrict_batch_mfma(int f32, const void* nullvecs, int nvec, const void* fine, void* coarse, int fhr, int fLy, int fnc, int chr, int cLy, int cnc, int bx, int by,
                        long fhalf_vol, long fsize, const int* ids8, int n, long cstride, long fstride, hipStream_t st) {
  if (!g_xfer_mfma || (g_xfer_mfma == 1 && !f32)) return SITE_DECLINED;
  // auto: where the matrix-core form is still the faster one AFTER the vector-FMA restricts stopped waiting for their loads one by one (qmg_common.h,
  // RawC): complex<float>, 5-8 systems, from the nc = 2 fine level to >= 16 null vectors (2048^2 -> 512^2 x 24: 1.09 against 1.33 ms).  Elsewhere
  // the vector kernels are equal or better now (512^2 -> 128^2 x 24, 8 systems: 0.55 against 0.58 ms; 1024^2 -> 256^2 x 8: 0.29 against 0.34;
  // every shape at <= 4 systems: 0.17-0.72 against 0.33-1.03 ms; profiles/r03_xfer_mfma.txt).
  if (g_xfer_mfma == 1 && !(n >= 5 && fnc <= 2 && nvec >= 16)) return SITE_DECLINED;
  XferGeomM g = {fhr, fLy, fnc, chr, cLy, cnc, bx, by, fhalf_vol, fsize};
  PassIdsM ids;
  ids.n = n;
  for (int q = 0; q < 8; q++) ids.id[q] = ids8[q];
  return f32 ? restrict_mfma_t<float>(nullvecs, nvec, fine, coarse, g, ids, cstride, fstride, st)
             : restrict_mfma_t<double>(nullvecs, nvec, fine, coarse, g, ids, cstride, fstride, st);
}
int prolong_batch_mfma(int f32, const void* nullvecs, int nvec, const void* coarse, void* fine, int fhr, int fLy, int fnc, int chr, int cLy, int cnc, int bx, int by,
                       long fhalf_vol, long fsize, const int* ids8, int n, long cstride, long fstride, hipStream_t st) {
  if (g_xfer_mfma < 2) return SITE_DECLINED;
  XferGeomM g = {fhr, fLy, fnc, chr, cLy, cnc, bx, by, fhalf_vol, fsize};
  PassIdsM ids;
  ids.n = n;
  for (int q = 0; q < 8; q++) ids.id[q] = ids8[q];
  return f32 ? prolong_mfma_t<float>(nullvecs, nvec, coarse, fine, g, ids, cstride, fstride, st)
             : prolong_mfma_t<double>(nullvecs, nvec, coarse, fine, g, ids, cstride, fstride, st);
}

}  // namespace qmg
